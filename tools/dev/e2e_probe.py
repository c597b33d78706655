#!/usr/bin/env python3
"""Dev probe: wall time of open(path) + one MappingQC -wgs job with the H2D of the compressed image in the background (pieces + events) vs in front.
usage: e2e_probe.py [reads=96000000] [variants=async4,async8,async2,sync]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
ngsqc = importlib.import_module("ngs-bits_amd")
import bamgen_lib as G
import hostprep as H
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 96_000_000
names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["async4", "async8", "async2", "sync", "async4"]
d = "/dev/shm" if os.path.isdir("/dev/shm") else os.environ.get("TMPDIR", "/tmp")
path = os.path.join(d, f"ngsqc_e2e_{reads}.bam")
t0 = time.time()
if not os.path.exists(path):
    G.generate(reads).tofile(path)
print(f"[e2e] {reads} reads, {os.path.getsize(path) / 1e9:.2f} GB at {path} in {time.time() - t0:.1f} s", flush=True)
omim = os.path.join(ROOT, "ngs-bits_amd", "resources", "hg38_440_omim_genes.bed")
for name in names:
    for k in list(os.environ):
        if k.startswith("NGSQC_") and k != "NGSQC_DEBUG":
            del os.environ[k]
    if name == "sync":
        os.environ["NGSQC_ASYNC_H2D"] = "0"
    elif name.startswith("noplan"):
        os.environ["NGSQC_ASYNC_PLAN"] = "0"; os.environ["NGSQC_H2D_THREADS"] = name[6:] or "4"
    else:
        os.environ["NGSQC_H2D_THREADS"] = name[5:]
    t0 = time.perf_counter()
    h = ngsqc.Handle(path=path, device=0)
    t1 = time.perf_counter()
    refs = h.refs
    regs, _ = H.bed_regions(omim, refs, 3); tx, ty = H.xy_tids(refs); sites = H.known_sites(refs)
    mp = dict(mode=ngsqc.MODE_WGS, regions=regs, min_mapq=1, tid_x=tx, tid_y=ty, nonspecial=H.nonspecial(refs))
    t2 = time.perf_counter()
    out = h.run_job(mapping=mp, sites=sites)
    t3 = time.perf_counter()
    h.upload_wait(); tm = h.timings()
    h.drop_decoded(); t4 = time.perf_counter(); h.run_job(mapping=mp, sites=sites); t5 = time.perf_counter()
    print(f"[e2e] {name:8s} open {t1 - t0:.3f} s, host prep {t2 - t1:.3f} s, first job {t3 - t2:.3f} s, open+job {t3 - t0 - (t2 - t1):.3f} s  ({tm['n_records'] / (t3 - t0 - (t2 - t1)) / 1e6:.1f} Mreads/s incl. H2D); "
          f"h2d {tm['h2d_ms']:.0f} ms = {os.path.getsize(path) / max(tm['h2d_ms'], 1e-9) / 1e6:.1f} GB/s; resident job {t5 - t4:.3f} s", flush=True)
    h.close()
os.remove(path)
