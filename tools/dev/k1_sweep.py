import sys, os, importlib
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
import bamgen_lib as G
ngsqc=importlib.import_module('ngs-bits_amd')
n=int(sys.argv[1]) if len(sys.argv)>1 else 24_000_000
# combos: park:p2variant:pipeline:sorted
combos=[tuple(int(y) for y in x.split(':')) for x in (sys.argv[2] if len(sys.argv)>2 else '16:2:1:0,16:2:1:1,16:2:0:0,16:2:0:1').split(',')]
img=G.generate(n, seed=11)
ref=None
h=ngsqc.Handle(data=img)
for park,p2,pipe,srt in combos:
    os.environ['NGSQC_P1_PARK']=str(park); os.environ['NGSQC_P2_VARIANT']=str(p2); os.environ['NGSQC_K1_PIPELINE']=str(pipe); os.environ['NGSQC_K1_SORTED']=str(srt)
    bh=bl=bi=1e9
    for it in range(3):
        h.drop_decoded(); h.decode()
        tm=h.timings(); bh=min(bh,tm['inflate_huff_ms']); bl=min(bl,tm['inflate_lz77_ms']); bi=min(bi,tm['inflate_ms'])
    out=h.inflated()
    if ref is None: ref=out
    print(f"park {park} p2 {p2} pipe {pipe} sorted {srt}: stage {bi:.2f} ms  huff {bh:.2f} ms ({tm['inflate_huff_launches']} launches)  lz77 {bl:.2f} ms  same_output={np.array_equal(out,ref)}", flush=True)
h.close()
