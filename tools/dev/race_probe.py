"""dev (next round, needs a GPU): the first job of a handle RACES the background H2D copy - the member table is walked in pieces (NGSQC_WALK_THREADS) so that
ngsqc_open returns early, the copy is slowed down (small pieces, a delay per piece) so that K1 chunks really wait for their pieces; counters must equal the
ones of a handle whose image is resident. usage: race_probe.py [reads=48000000]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
ngsqc = importlib.import_module("ngs-bits_amd")
import bamgen_lib as G  # noqa: E402
import hostprep as H  # noqa: E402

OMIM = os.path.join(ROOT, "ngs-bits_amd", "resources", "hg38_440_omim_genes.bed")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48_000_000
path = f"/dev/shm/ngsqc_race_{n}.bam"
G.generate(n).tofile(path)
try:
    def job(env):
        for k in ("NGSQC_WALK_THREADS", "NGSQC_H2D_PIECE_MB", "NGSQC_H2D_DELAY_US", "NGSQC_H2D_THREADS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        t0 = time.perf_counter(); h = ngsqc.Handle(path=path); t1 = time.perf_counter()
        refs = h.refs; regs, _ = H.bed_regions(OMIM, refs, 3); tx, ty = H.xy_tids(refs)
        mp = dict(mode=ngsqc.MODE_WGS, regions=regs, min_mapq=1, tid_x=tx, tid_y=ty, nonspecial=H.nonspecial(refs))
        out = h.run_job(mapping=mp, sites=H.known_sites(refs))
        t2 = time.perf_counter(); h.upload_wait(); tm = h.timings(); h.close()
        return out, t1 - t0, t2 - t0, tm["h2d_ms"]
    ref, o0, j0, h0 = job({})
    print(f"[race] default: open {o0:.3f} s, open + job {j0:.3f} s, h2d {h0:.0f} ms")
    for env in ({"NGSQC_WALK_THREADS": "8"}, {"NGSQC_WALK_THREADS": "8", "NGSQC_H2D_PIECE_MB": "8", "NGSQC_H2D_DELAY_US": "2000", "NGSQC_H2D_THREADS": "2"}):
        out, o, j, hh = job(env)
        same = np.array_equal(np.asarray(out["counters"]), np.asarray(ref["counters"]))
        print(f"[race] {env}: open {o:.3f} s, open + job {j:.3f} s, h2d {hh:.0f} ms, same counters {same}")
finally:
    os.remove(path)
