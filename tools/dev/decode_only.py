import sys, os, importlib
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import bamgen_lib as G
ngsqc=importlib.import_module('ngs-bits_amd')
n=int(sys.argv[1]) if len(sys.argv)>1 else 8_000_000
img=G.generate(n, seed=11)
h=ngsqc.Handle(data=img)
for it in range(2):
    h.drop_decoded(); h.decode()
print(h.timings())
