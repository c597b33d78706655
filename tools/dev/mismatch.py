import sys, importlib
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, oracle_lib as O
ngsqc=importlib.import_module('ngs-bits_amd')
p='tests/golden/ref_in/close_exons.bam'
ob=O.Bam(p); ref=ob.inflated()
L=ngsqc.lib()
import ctypes as C
# open may fail at header parse; use open_memory on raw and catch
try:
    h=ngsqc.Handle(path=p); got=h.inflated()
except Exception as e:
    print('open failed', e); sys.exit()
d=np.nonzero(got!=ref)[0]
print('mismatches', d.size, 'first', d[:20])
i=int(d[0]); print(bytes(ref[i-40:i+40])); print(bytes(got[i-40:i+40]))
