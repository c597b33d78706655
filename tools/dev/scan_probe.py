"""Round-5 probe of the scan stage (K2-K6 on resident inflated data, un-pipelined): one generated BAM, the fused job under several settings of the
chain walk (NGSQC_WALKERS, NGSQC_WALK_WAVES, ...), every result compared with the first. Prints one line per setting:
  t_scan = index_ms + scan_ms + finalize_ms of an NGSQC_PIPELINE=0 step, the share of the HBM roofline (sum of 4 + block_size over the records / t_scan / 8 TB/s).
usage: python tools/dev/scan_probe.py [--reads N] [--ont] [--tool mappingqc|bedcoverage|bedlowcoverage] [--reps R] [--sets "A=1,B=2;C=3;..."]"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=48_000_000)
    ap.add_argument("--ont", action="store_true")
    ap.add_argument("--tool", default="mappingqc")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--min-baseq", type=int, default=0)
    ap.add_argument("--sets", default="NGSQC_WALKERS=1;NGSQC_WALKERS=2;NGSQC_WALKERS=4;NGSQC_WALKERS=8;NGSQC_WALKERS=4,NGSQC_WALK_WAVES=4;NGSQC_WALKERS=8,NGSQC_WALK_WAVES=4")
    ap.add_argument("--pipelined", action="store_true", help="also time pipelined steps (job wall)")
    ap.add_argument("--no-default", action="store_true", help="only the settings of --sets")
    ap.add_argument("--image-cache", default=None, help="keep the generated BAM in this file (profiling: several runs of one command)")
    args = ap.parse_args()
    ngsqc = importlib.import_module("ngs-bits_amd")
    import bamgen_lib as G
    import hostprep as H
    sys.path.insert(0, ROOT)
    import bench as B

    t0 = time.time()
    if args.image_cache and os.path.exists(args.image_cache):
        image = np.fromfile(args.image_cache, dtype=np.uint8)
    else:
        image = G.generate(args.reads, seed=20260821, mode=1 if args.ont else 0, depth=40.0 if args.ont else 30.0, level=6, aligned=True)
        if args.image_cache:
            image.tofile(args.image_cache)
    print(f"[probe] generated {args.reads} reads, {image.size / 1e9:.2f} GB in {time.time() - t0:.1f} s", flush=True)
    h = ngsqc.Handle(data=image, device=0)
    refs = h.refs
    tx, ty = H.xy_tids(refs)
    sites = H.known_sites(refs)
    if args.tool == "mappingqc":
        regs, _ = H.bed_regions(os.path.join(ROOT, "ngs-bits_amd", "resources", "hg38_440_omim_genes.bed"), refs, 3)
        mp = dict(mode=ngsqc.MODE_WGS, regions=regs, min_mapq=1, tid_x=tx, tid_y=ty, nonspecial=H.nonspecial(refs))

        def step():
            h.drop_decoded()
            out = h.run_job(mapping=mp, sites=sites, site_params=(1, 13, args.ont))
            c = out["counters"]
            roi_bases = int(c[26]); usable_roi = int(c[14])
            half = int(round(0.5 * usable_roi / roi_bases)) if roi_bases else 0
            hist, cov = h.depth_stats(599, half)
            return np.concatenate([np.asarray(c, dtype=np.int64), np.asarray(hist, dtype=np.int64), np.asarray(out["site_counts"], dtype=np.int64).ravel()])
    else:
        bed = os.path.join(os.environ.get("TMPDIR", "/tmp"), "probe_exome.bed")
        B.synthetic_exome_bed(bed, refs, 20260821, overlapping=(args.tool == "bedcoverage"))
        union, _ = H.bed_regions(bed, refs, 2); lines, _ = H.bed_regions(bed, refs, 0)
        union_c = ngsqc.capi._regions_array(np.array(union, dtype=np.int32)); lines_c = ngsqc.capi._regions_array(np.array(lines, dtype=np.int32))

        def step():
            h.drop_decoded()
            if args.tool == "bedcoverage":
                h.scan_depth(union_c, min_mapq=1, n_regions=len(union))
                return np.asarray(h.region_sums(lines_c, n_lines=len(lines)), dtype=np.int64)
            h.scan_depth(union_c, min_mapq=1, min_baseq=args.min_baseq, n_regions=len(union))
            r = h.lowhigh_runs(lines_c, 20, is_high=False, saturate254=True, n_lines=len(lines), as_array=True)
            return np.frombuffer(bytes(r), dtype=np.uint8).copy()

    ref = None
    for spec in ([] if args.no_default else [""]) + [x for x in args.sets.split(";") if x]:
        env = dict(kv.split("=") for kv in spec.split(",") if kv)
        for k, v in env.items():
            os.environ[k] = v
        try:
            step()   # warm-up (buffers)
            os.environ["NGSQC_PIPELINE"] = "0"
            best = None
            for _ in range(args.reps):
                res = step(); tm = h.timings()
                t_scan = tm["index_ms"] + tm["scan_ms"] + tm["finalize_ms"]
                if best is None or t_scan < best[0]:
                    best = (t_scan, tm)
            del os.environ["NGSQC_PIPELINE"]
            wall = None
            if args.pipelined:
                ts = time.perf_counter()
                for _ in range(args.reps):
                    step()
                wall = (time.perf_counter() - ts) / args.reps * 1e3
            t_scan, tm = best
            ok = True
            if ref is None:
                ref = res
            else:
                ok = res.shape == ref.shape and bool(np.array_equal(res, ref))
            frac = tm["scan_algorithmic_bytes"] / max(t_scan, 1e-9) / 1e6 / 8000.0
            kfrac = tm["scan_algorithmic_bytes"] / max(tm["scan_kernel_ms"], 1e-9) / 1e6 / 8000.0
            print(f"[probe] {spec or 'default':48s} t_scan {t_scan:8.3f} ms (index {tm['index_ms']:.3f} scan {tm['scan_ms']:.3f} kernels {tm['scan_kernel_ms']:.3f} finalize {tm['finalize_ms']:.3f} pileup {tm['pileup_ms']:.3f}) "
                  f"frac {frac:.3f} kernel-only {kfrac:.3f} tiles {tm['n_tiles']} on-device {tm['tiles_chain_on_device']} fused {tm['tiles_scan_fused']} walkers {tm['walkers_per_member']} records {tm['n_records']}"
                  + (f" job wall {wall:.1f} ms = {tm['n_records'] / wall / 1e3:.1f} Mreads/s" if wall else "") + f" same_result {ok}", flush=True)
        finally:
            for k in env:
                os.environ.pop(k, None)
            os.environ.pop("NGSQC_PIPELINE", None)
    h.close()


if __name__ == "__main__":
    main()
