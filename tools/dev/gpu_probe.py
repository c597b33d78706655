import sys, time, importlib
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import bamgen_lib as G, hostprep as H
ngsqc=importlib.import_module('ngs-bits_amd')
import os
print('cores', os.cpu_count())
t=time.time(); img=G.generate(4_000_000, seed=7); print('gen s', time.time()-t, 'bytes', img.size)
t=time.time(); h=ngsqc.Handle(data=img); print('open s', time.time()-t)
regs,_=H.bed_regions('ngs-bits_amd/resources/hg38_440_omim_genes.bed', h.refs, 3)
tx,ty=H.xy_tids(h.refs)
for it in range(3):
    h.drop_decoded()
    t=time.time(); c,_=h.scan_mapping(ngsqc.MODE_WGS, regions=regs, tid_x=tx, tid_y=ty, nonspecial=H.nonspecial(h.refs)); w=time.time()-t
    tm=h.timings(); print('wall', w, tm)
print('reads/s end-to-end', tm['n_records']/ (tm['total_ms']/1e3)/1e6, 'M; inflate GB/s out', tm['inflated_bytes']/tm['inflate_ms']/1e6, 'scan GB/s', tm['scan_algorithmic_bytes']/tm['scan_ms']/1e6)
