import sys, os, time, importlib
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
import bamgen_lib as G, hostprep as H
ngsqc=importlib.import_module('ngs-bits_amd')
n=int(sys.argv[1]) if len(sys.argv)>1 else 150_000
t=time.time(); img=G.generate(n, seed=5, mode=1, depth=40.0); print('gen', round(time.time()-t,1),'s', img.size/1e9,'GB compressed', flush=True)
h=ngsqc.Handle(data=img)
regs,_=H.bed_regions(os.path.join('ngs-bits_amd','resources','hg38_440_omim_genes.bed'), h.refs, 3)
tx,ty=H.xy_tids(h.refs); ns=H.nonspecial(h.refs)
for it in range(2):
    h.drop_decoded(); t0=time.perf_counter()
    c,_=h.scan_mapping(ngsqc.MODE_WGS, regions=regs, tid_x=tx, tid_y=ty, nonspecial=ns)
    t1=time.perf_counter(); tm=h.timings()
    rq=h.scan_reads(True); t2=time.perf_counter()
    sites=np.array([(0,p,p) for p in range(1_000_000, 200_000_000, 30_000)],dtype=np.int32)
    pc=h.site_pileup(sites,1,13,True); t3=time.perf_counter()
    print(f"records {tm['n_records']} inflated {tm['inflated_bytes']/1e9:.2f} GB | mapping {1e3*(t1-t0):.1f} ms (inflate {tm['inflate_ms']:.1f} index {tm['index_ms']:.1f} scan {tm['scan_ms']:.1f} [kern {tm['scan_kernel_ms']:.1f}]) | reads {1e3*(t2-t1):.1f} ms | pileup {1e3*(t3-t2):.1f} ms | max_len {rq['max_cycles']} bases {rq['bases_sequenced']/1e9:.2f} G", flush=True)
