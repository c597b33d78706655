import sys, os, time, importlib
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import bamgen_lib as G, hostprep as H
ngsqc=importlib.import_module('ngs-bits_amd')
img=G.generate(24_000_000, seed=20260821)
h=ngsqc.Handle(data=img)
regs,_=H.bed_regions('ngs-bits_amd/resources/hg38_440_omim_genes.bed', h.refs, 3)
tx,ty=H.xy_tids(h.refs); ns=H.nonspecial(h.refs)
for it in range(12):
    t0=time.perf_counter(); h.drop_decoded(); t1=time.perf_counter()
    c,_=h.scan_mapping(ngsqc.MODE_WGS, regions=regs, tid_x=tx, tid_y=ty, nonspecial=ns); t2=time.perf_counter()
    hist,cov=h.depth_stats(599, 15); t3=time.perf_counter()
    tm=h.timings()
    print(f"step {it}: drop {1e3*(t1-t0):.1f} scan_mapping {1e3*(t2-t1):.1f} depth_stats {1e3*(t3-t2):.1f} | inflate {tm['inflate_ms']:.1f} index {tm['index_ms']:.1f} total {tm['total_ms']:.1f}", flush=True)
