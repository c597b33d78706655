import sys, os, time, importlib
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
import bamgen_lib as G
ngsqc=importlib.import_module('ngs-bits_amd')
n=int(sys.argv[1]) if len(sys.argv)>1 else 8_000_000
variants=[int(x) for x in sys.argv[2].split(',')] if len(sys.argv)>2 else [0,1,2,3,4,5,6]
img=G.generate(n, seed=11)
ref=None
for v in variants:
    os.environ['NGSQC_INFLATE_VARIANT']='0'
    h=ngsqc.Handle(data=img)
    os.environ['NGSQC_INFLATE_VARIANT']=str(v)
    best=1e9
    for it in range(3):
        h.drop_decoded()
        try: h.decode()
        except Exception as e: pass
        tm=h.timings(); best=min(best, tm['inflate_ms'])
    try: out=h.inflated()
    except Exception: out=np.zeros(1,np.uint8)
    if ref is None: ref=out
    ok=np.array_equal(out,ref)
    print(f"variant {v}: inflate {best:.2f} ms  out {tm['inflated_bytes']/best/1e6:.1f} GB/s  in {tm['compressed_bytes']/best/1e6:.1f} GB/s  same_output={ok}", flush=True)
    h.close()
