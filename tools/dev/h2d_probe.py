#!/usr/bin/env python3
"""Dev probe: H2D rate of ngsqc_open_memory for a ~14 GB image under NGSQC_H2D_THREADS = 1 (one hipMemcpy of pageable memory) / 4 / 8 / 16."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
ngsqc = importlib.import_module("ngs-bits_amd")
import bamgen_lib as G
img = G.generate(48_000_000)
body = img[:-28]
big = np.concatenate([body, body, body, img[-28:]])   # (valid BGZF, header members first: enough for open; the records are never decoded here)
print(f"[h2d] image {big.size / 1e9:.2f} GB", flush=True)
for T in (1, 8, 4, 16, 1, 8):
    os.environ["NGSQC_H2D_THREADS"] = str(T)
    t0 = time.time(); h = ngsqc.Handle(data=big, device=0); dt = time.time() - t0
    print(f"[h2d] threads {T:2d}: open {dt:.2f} s, h2d {h.timings()['h2d_ms']:.0f} ms = {big.size / h.timings()['h2d_ms'] / 1e6:.1f} GB/s", flush=True)
    h.close()
