#!/usr/bin/env python3
"""bench.py — the MappingQC / coverage hot path on MI355X (BASELINE.json metric: Mreads/s + achieved HBM GB/s).

Default workload = BASELINE.json configs[1]: MappingQC -wgs on a synthetic 30x WGS BAM (~6.2e8 2x150 bp PE reads, coordinate-sorted,
zlib-6 BGZF, SURVEY.md §8(d) config 2 statistics; ~60 GB compressed, ~209 GB inflated) on one MI355X. The compressed image is resident
in HBM when the timed region starts (the H2D time of the image is reported separately, and an end-to-end rate next to `value`).
A "step" is what `bin/MappingQC -wgs` does with the BAM, as ONE fused job (ngsqc_run_job): every BGZF member is inflated once (K1),
records are indexed (K2), and the mapping_wgs scan (counters, insert-size histogram, chrX/chrY counts, OMIM-ROI depth scatter) plus the
contamination pileup of the known common SNVs see every tile of the inflated stream while it is resident; then K6 (depth prefix sum,
histogram, half-depth count) and the result copies. The file is larger than HBM once inflated, so it streams through three tile buffers
(K1 of tile t+1 overlaps K2 + consumers of tile t). If the host cannot hold the full image the read count is scaled down and
`config.workload` says so.

  --gpus N     one process per GPU (spawned here when no launcher set WORLD_SIZE), one BAM per GPU (weak scaling), RCCL all-reduce of the
               counter vectors at the end of every step
  --tool       mappingqc (default) | bedcoverage | bedlowcoverage   (configs[2]: exome-shaped BED over the same BAM)
  --ont        configs[4]: ONT-like long reads (N50 ~20 kb, CG-tag records), MappingQC -wgs -single_end shape

Prints ONE JSON line (rank 0). Extra objects: roofline (dominant kernel by time), roofline_k1_stage, roofline_scan (SURVEY.md §8(d):
t_scan = K2-K6 on resident inflated data, taken from one extra un-pipelined step), stage_ms, end_to_end, cpu_baseline (oracle, 1 thread,
bounded sample), cpu_baseline_all_cores (same loop on every host core over the whole file; its additive counters and depth histogram are
the parity check of the bench input: counters_match_gpu).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
FULL_READS = 620_000_000   # 30x of hg38 at 150 bp
BYTES_PER_READ_COMPRESSED = 98   # of the synthetic BAM (zlib-6)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=0, help="records per GPU per step (0 = the full 30x file when the host can hold it)")
    ap.add_argument("--seed", type=int, default=20260821)
    ap.add_argument("--tool", default="mappingqc", choices=["mappingqc", "bedcoverage", "bedlowcoverage"])
    ap.add_argument("--min-baseq", type=int, default=0, help="bedlowcoverage: -min_baseq")
    ap.add_argument("--ont", action="store_true", help="configs[4]: ONT-like long reads")
    ap.add_argument("--flavor", type=int, default=0, help="short-read generator: 0 = SURVEY.md 8(d) shape (random SEQ, 4-level QUAL); bit 0: SEQ from a synthetic reference (overlapping reads share "
                    "sequence); bits 1-2: QUAL model 1 = eight bins, 2 = forty levels")
    ap.add_argument("--level", type=int, default=6, help="zlib level of the generated BGZF members")
    ap.add_argument("--cpu-sample-reads", type=int, default=12_000_000, help="records of the same batch timed on one host core")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU smoke tests of the N>1 path)")
    ap.add_argument("--single-bam", action="store_true", help="strong scaling: ONE BAM sharded over the ranks by BGZF member range "
                    "(all-gather of shard summaries, SUM all-reduce of counters and of the int32 difference array) instead of one BAM per rank")
    ap.add_argument("--image-cache", default="", help="keep / reuse the generated BAM image at this path (profiling passes on one box)")
    ap.add_argument("--all-ranks-on-device0", action="store_true", help="debug: map every rank to cuda:0")
    return ap.parse_args()


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start one process per GPU here (rank 0 prints the JSON line)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, p.wait())
    return rc


def host_memory_available():
    try:
        import psutil
        return int(psutil.virtual_memory().available)
    except Exception:
        return 0


def synthetic_exome_bed(path, refs, seed, n_intervals=200_000, overlapping=False):
    """exon-like intervals, length ~ LogNormal(5.0, 0.6) clipped to [60, 2000], spread over chr1-22,X,Y in
    proportion to their length (sum ~38 Mb); overlapping=True adds 10 % lines that overlap their predecessor (BedCoverage's unmerged input)."""
    rng = np.random.default_rng(seed)
    chroms = [(n, l) for n, l in refs if n not in ("chrM", "MT")][:24]
    tot = float(sum(l for _, l in chroms))
    lines = []
    for name, ln in chroms:
        k = int(round(n_intervals * ln / tot))
        lens = np.clip(np.exp(rng.normal(5.0, 0.6, k)), 60, 2000).astype(np.int64)
        starts = np.sort(rng.integers(1000, ln - 3000, k))
        # keep them disjoint: push every start behind the previous end
        s = starts.copy(); prev_end = 0
        for i in range(k):
            if s[i] <= prev_end + 1:
                s[i] = prev_end + 2
            prev_end = s[i] + lens[i]
            if prev_end >= ln - 10:
                s = s[:i]; lens = lens[:i]; break
        for i in range(len(s)):
            lines.append((name, int(s[i]), int(s[i] + lens[i])))
            if overlapping and rng.random() < 0.10:
                lines.append((name, int(s[i] + lens[i] // 2), int(s[i] + lens[i] + 30)))
    with open(path, "w") as f:
        for c, a, b in lines:
            f.write(f"{c}\t{a}\t{b}\n")
    return len(lines)


def prefix_image(image, target_bytes):
    """The BGZF members of the image's first ~target_bytes plus the EOF marker: a valid BAM with the first records of the batch (aligned input:
    members end at record ends)."""
    import struct
    pos = 0; n = image.size
    while pos < min(target_bytes, n - 28):
        pos += struct.unpack_from("<H", image, pos + 16)[0] + 1
    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    out = np.empty(pos + 28, dtype=np.uint8)
    out[:pos] = image[:pos]; out[pos:] = np.frombuffer(eof, dtype=np.uint8)
    return out



def scan_roofline(tm):
    """t_scan = K2-K6 on resident inflated data (an un-pipelined step's HIP-event stage times do not overlap)"""
    t_scan = tm["index_ms"] + tm["scan_ms"] + tm["finalize_ms"]
    b = int(tm["scan_algorithmic_bytes"])
    gbs = b / max(t_scan, 1e-9) / 1e6
    return {"stage": "K2-K6 on resident inflated data, un-pipelined step", "bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 5), "algorithmic_bytes": b, "t_scan_ms": round(t_scan, 4),
            "scan_kernel_only": {"frac": round(b / max(tm["scan_kernel_ms"], 1e-9) / 1e6 / HBM_PEAK_GBS, 5), "ms": round(tm["scan_kernel_ms"], 4)},
            "itemised_ms": {"k2_index_incl_start_guess": round(tm["index_ms"], 4), "scan_kernels": round(tm["scan_kernel_ms"], 4),
                            "scan_stage_other": round(tm["scan_ms"] - tm["scan_kernel_ms"], 4), "depth_finalize": round(tm["finalize_ms"], 4)}}


def timed_steps(step, h, steps):
    """one warm-up, `steps` timed whole-job steps (compressed image resident, every step from the compressed bytes), one extra un-pipelined step for the stage times"""
    import torch
    step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    tm = h.timings()
    os.environ["NGSQC_PIPELINE"] = "0"
    try:
        step(); serial = h.timings()
    finally:
        del os.environ["NGSQC_PIPELINE"]
    n_rec = int(tm["n_records"])
    return res, {"value": round(n_rec * steps / el / 1e6, 3), "unit": "Mreads/s", "steps": steps, "ms_per_step": round(el / steps * 1e3, 3), "reads_per_step": n_rec,
                 "members_inflated_per_step": int(tm["members_inflated"]), "tiles": int(tm["n_tiles"]), "tiles_scan_fused": int(tm["tiles_scan_fused"]),
                 "walkers_per_member": int(tm["walkers_per_member"]), "roofline_scan": scan_roofline(serial), "inflate_stage_unpipelined_ms": round(serial["inflate_ms"], 3)}


def coverage_tool_leg(ngsqc, H, O, h, image, refs, tool, min_baseq, steps, args, device):
    """configs[2]: BedCoverage / BedLowCoverage over the SAME resident 30x image as the headline step (exome-shaped BED, 200 000 intervals,
    ~38 Mb; BedCoverage on the unmerged variant with 10 % overlapping lines; BedLowCoverage -cutoff 20 with and without -min_baseq 20). Reference:
    WorkerAverageCoverage.cpp:86-173, WorkerLowOrHighCoverage.cpp:141-252."""
    tag = tool + (f"_baseq{min_baseq}" if min_baseq else "")
    bed_path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"ngsqc_bench_exome_{args.seed}_{tag}.bed")
    n_lines = synthetic_exome_bed(bed_path, refs, args.seed, overlapping=(tool == "bedcoverage"))
    union, _ = H.bed_regions(bed_path, refs, 2); lines, _ = H.bed_regions(bed_path, refs, 0)
    union_c = ngsqc.capi._regions_array(np.array(union, dtype=np.int32)); lines_c = ngsqc.capi._regions_array(np.array(lines, dtype=np.int32))

    def step():
        h.drop_decoded()
        if tool == "bedcoverage":
            h.scan_depth(union_c, min_mapq=1, n_regions=len(union))
            return h.region_sums(lines_c, n_lines=len(lines))
        h.scan_depth(union_c, min_mapq=1, min_baseq=min_baseq, n_regions=len(union))
        return h.lowhigh_runs(lines_c, 20, is_high=False, saturate254=True, n_lines=len(lines), as_array=True)
    _, out = timed_steps(step, h, steps)
    out["workload"] = ({"bedcoverage": "BedCoverage: depth scan over the merged exome regions + per-line sums (unmerged BED, 10 % overlapping lines)",
                        "bedlowcoverage": f"BedLowCoverage -cutoff 20{' -min_baseq ' + str(min_baseq) if min_baseq else ''}: depth scan + low-coverage runs, sweep saturation"}[tool]
                       + f"; exome BED of {n_lines} lines, {int(sum(e - s + 1 for _, s, e in union))} merged bases; the full resident image, every step from the compressed bytes")
    if not args.no_cpu_baseline:
        # the oracle's restatement of the tool (1 thread) on the first records of the same BAM; the same sample through the GPU path is the parity check
        samp = prefix_image(image, min(int(image.size), args.cpu_sample_reads * BYTES_PER_READ_COMPRESSED))
        sp = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"ngsqc_bench_sample_{args.seed}_{tag}.bam")
        samp.tofile(sp)
        try:
            t1 = time.time()
            ob = O.Bam(sp)                                  # (sequential inflate + record framing of the sample: part of the CPU tool's work)
            t_load = time.time() - t1
            hs = ngsqc.Handle(data=samp, device=device)
            t1 = time.time()
            if tool == "bedcoverage":
                cov, _, _ = O.avg_coverage(ob, bed_path, merge_bed=False, min_mapq=1, random_access=False)
                secs = time.time() - t1 + t_load
                hs.scan_depth(union, min_mapq=1); ok = bool(np.array_equal(hs.region_sums(lines), cov))
            else:
                exp = O.low_high_coverage(ob, bed_path, 20, 1, min_baseq, is_high=False, random_access=False, tool_merge=1)
                secs = time.time() - t1 + t_load
                hs.scan_depth(union, min_mapq=1, min_baseq=min_baseq)
                ok = bool(np.array_equal(np.minimum(hs.depth(exp["roi_bases"]), 254), np.minimum(exp["depth"], 254)))
            hs.close()
            out["cpu_baseline"] = {"value": round(ob.count / secs / 1e6, 4), "unit": "Mreads/s", "cores": 1, "kind": "port",
                                   "sample": f"first {ob.count} records of the same BAM: sequential inflate + record framing + the oracle's restatement of the tool's sweep, 1 thread, {secs:.1f} s"}
            out["counters_match_gpu"] = ok
            out["counters_match_note"] = "per-line depth sums (BedCoverage) / per-base depth of the whole exome BED (BedLowCoverage) of the sample, GPU vs oracle, bit-exact"
        finally:
            os.remove(sp)
    os.remove(bed_path)
    return tag, out


def ont_leg(ngsqc, G, H, O, args, device, steps):
    """configs[4]: MappingQC -wgs (fused job incl. contamination pileup) on the largest piece of the 40x ONT file that fits the bench's time and the host's memory:
    2 000 000 of ~8e6 reads by default (N50 ~20 kb, ~1 CIGAR op per 12 bp, CG-tag records: ~22 GB compressed, ~84 GB inflated, ~2.5 min of generator; NGSQC_BENCH_ONT_READS
    overrides, the whole file is 8 000 000) - scaled down when the host cannot hold it. Reference: Statistics.cpp:1068-1182 over BamReader::cigarData (long CIGAR / CG tag)."""
    reads = int(os.environ.get("NGSQC_BENCH_ONT_READS", "2000000"))
    avail = host_memory_available()
    if avail and reads > 100_000:   # ~11 KB of compressed image per read, twice that while the generator's pieces are joined
        reads = max(100_000, min(reads, int(0.4 * avail / 11_000) // 100_000 * 100_000))
    gen_kw = dict(seed=args.seed, mode=1, depth=40.0, first_contig=0, start_pos=0, level=args.level, aligned=True, flavor=0)
    t0 = time.time(); image = G.generate(reads, threads=2 * G.effective_cpus(), **gen_kw); gen_s = time.time() - t0
    h = ngsqc.Handle(data=image, device=device)
    refs = h.refs
    omim = os.path.join(ROOT, "ngs-bits_amd", "resources", "hg38_440_omim_genes.bed")
    tx, ty = H.xy_tids(refs); regs, _ = H.bed_regions(omim, refs, 3); sites_arr = H.known_sites(refs)
    mp = dict(mode=ngsqc.MODE_WGS, regions=regs, min_mapq=1, tid_x=tx, tid_y=ty, nonspecial=H.nonspecial(refs))

    def step():
        h.drop_decoded()
        return h.run_job(mapping=mp, sites=sites_arr, site_params=(1, 13, True))["counters"]
    _, out = timed_steps(step, h, steps)
    tm = h.timings()
    out["workload"] = (f"MappingQC -wgs as one fused job on {out['reads_per_step']} of the ~8e6 reads of configs[4] (the 40x ONT-like BAM: N50 ~20 kb, ~1 CIGAR op per 12 bp, "
                       f"CG-tag records), {int(tm['compressed_bytes'])} compressed / {int(tm['inflated_bytes'])} inflated bytes in {int(tm['n_tiles'])} tiles; compressed image resident, "
                       f"every step from the compressed bytes")
    out["gbases_per_s"] = round(out["value"] * 1e6 * (int(tm["inflated_bytes"]) / max(out["reads_per_step"], 1)) / 1.72 / 1e9, 2)   # (~1.72 inflated bytes per base: 4-bit SEQ + QUAL + CIGAR share)
    out["generate_s"] = round(gen_s, 1)
    h.close()
    if not args.no_cpu_baseline:
        c_cpu, st, secs = O.baseline_wgs_stream(image, omim, 1, min(150_000, reads), sites=sites_arr, site_params=(1, 13, True))
        out["cpu_baseline"] = {"value": round(st["n_records"] / secs / 1e6, 5), "unit": "Mreads/s", "cores": 1, "kind": "port",
                               "sample": f"first {st['n_records']} records of the same BAM ({st['compressed']} compressed bytes), oracle/stream.hpp single-thread sequential loop (the same job incl. the site pileup), {secs:.1f} s"}
        del image
        # long reads span BGZF members, so the file cannot be cut into per-thread member ranges for an all-cores parity pass: parity of the generator's data is
        # checked on a 200 000-read BAM of the same generator (all counters of the GPU job vs the sequential oracle, bit-exact)
        n_par = min(reads, int(os.environ.get("NGSQC_BENCH_ONT_PARITY_READS", "200000")))
        small = G.generate(n_par, **dict(gen_kw, threads=2 * G.effective_cpus()))
        hs = ngsqc.Handle(data=small, device=device)
        got = hs.run_job(mapping=mp)["counters"]; n_tiles_small = int(hs.timings()["n_tiles"]); hs.close()
        t1 = time.time(); c_small, _, _ = O.baseline_wgs_stream(small, omim, 1, -1); t_par = time.time() - t1
        out["counters_match_gpu"] = bool(all(int(got[i]) == int(c_small[i]) for i in range(len(got)) if i not in (27, 28)))
        out["counters_match_note"] = f"a {n_par}-read BAM of the same generator and seed ({n_tiles_small} tiles on the device): all counters of the GPU job vs the sequential oracle ({t_par:.0f} s), bit-exact"
    return out


def flavor_leg(ngsqc, G, H, O, args, device, steps, flavor=5, reads=None):
    """K1's cost is the token mix: the headline is measured on SURVEY.md 8(d)'s shape (random SEQ, 4-level QUAL: ratio 3.45, matches of 11 bytes on average), the easiest
    one for phase 2. This leg runs the same fused MappingQC -wgs job on a 96 M-read shard of generator flavor 5 (SEQ from a synthetic reference genome - overlapping reads
    share sequence, as real data does - and 40-level qualities: ratio 2.5, literal-heavy). No reference line; SURVEY 8(d) config 2 is the spec of the shape."""
    if reads is None:
        reads = int(os.environ.get("NGSQC_BENCH_FLAVOR_READS", "96000000"))
    gen_kw = dict(seed=args.seed, mode=0, depth=30.0, first_contig=0, start_pos=0, level=args.level, aligned=True, flavor=flavor)
    t0 = time.time(); image = G.generate(reads, threads=2 * G.effective_cpus(), **gen_kw); gen_s = time.time() - t0
    h = ngsqc.Handle(data=image, device=device)
    refs = h.refs
    omim = os.path.join(ROOT, "ngs-bits_amd", "resources", "hg38_440_omim_genes.bed")
    tx, ty = H.xy_tids(refs); regs, _ = H.bed_regions(omim, refs, 3); sites_arr = H.known_sites(refs)
    mp = dict(mode=ngsqc.MODE_WGS, regions=regs, min_mapq=1, tid_x=tx, tid_y=ty, nonspecial=H.nonspecial(refs))
    last = {}

    def step():
        h.drop_decoded()
        last["c"] = h.run_job(mapping=mp, sites=sites_arr, site_params=(1, 13, False))["counters"]
        return last["c"]
    _, out = timed_steps(step, h, steps)
    tm = h.timings()
    out["workload"] = (f"MappingQC -wgs as one fused job on a {out['reads_per_step']}-read shard of generator flavor {flavor} (SEQ from a synthetic reference genome, 40-level QUAL; "
                       f"{int(tm['compressed_bytes'])} compressed / {int(tm['inflated_bytes'])} inflated bytes, ratio {int(tm['inflated_bytes']) / max(int(tm['compressed_bytes']), 1):.2f})")
    out["generate_s"] = round(gen_s, 1)
    got = np.asarray(last["c"]).copy()
    h.close()
    if not args.no_cpu_baseline:
        # parity on a prefix of the same image: the sequential oracle against the GPU job over the same records
        samp = prefix_image(image, min(int(image.size), 6_000_000 * BYTES_PER_READ_COMPRESSED))
        hs = ngsqc.Handle(data=samp, device=device)
        g2 = hs.run_job(mapping=mp, sites=sites_arr, site_params=(1, 13, False))["counters"]; hs.close()
        t1 = time.time(); c_cpu, st, secs = O.baseline_wgs_stream(samp, omim, 1, -1, sites=sites_arr, site_params=(1, 13, False))
        out["cpu_baseline"] = {"value": round(st["n_records"] / secs / 1e6, 4), "unit": "Mreads/s", "cores": 1, "kind": "port",
                               "sample": f"the first {st['n_records']} records of the same BAM, oracle/stream.hpp single-thread sequential loop, {secs:.1f} s"}
        out["counters_match_gpu"] = bool(all(int(g2[i]) == int(c_cpu[i]) for i in range(len(g2)) if i not in (27, 28)))
        out["counters_match_note"] = "all counters of the GPU job over the sample vs the sequential oracle over the same records, bit-exact"
    del got
    return out


def main():
    args = parse_args()
    t_main = time.time(); marks = []   # (phase, seconds since start): the run's wall-time budget, printed to stderr and kept in the line ("budget")

    def mark(what):
        marks.append((what, round(time.time() - t_main, 1)))
        if int(os.environ.get("RANK", "0")) == 0:
            print(f"[bench] t+{marks[-1][1]:.1f} s {what}", file=sys.stderr, flush=True)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import datetime
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world, timeout=datetime.timedelta(hours=2))   # (rank 0 generates the BAM for minutes while the others wait)
        assert dist.get_world_size() == world
    if args.gpus != world and world > 1 and rank == 0:
        print(f"[bench] --gpus {args.gpus} but the launcher started {world} ranks: using {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.all_ranks_on_device0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if args.backend == "nccl" else torch.device("cpu")

    ngsqc = importlib.import_module("ngs-bits_amd")
    import bamgen_lib as G
    import hostprep as H

    # the product's own collective (include/ngsqc.h ngsqc_comm_*: RCCL loaded by libngsqc_hip.so) carries every exchange of the data path; torch.distributed only
    # starts the ranks, hands the 128-byte unique id around and provides the timing barriers of the bench contract
    comm = None; comm_note = None
    if world > 1 and args.backend == "nccl" and not args.all_ranks_on_device0 and os.environ.get("NGSQC_BENCH_TORCH_COLLECTIVE") is None:
        # (every rank must end up on the same path: a rank whose communicator could not be made sends the others back to torch.distributed)
        box = [None]
        if rank == 0:
            try:
                box[0] = ngsqc.Comm.unique_id()
            except Exception as e:
                comm_note = f"ngsqc_comm_unique_id failed: {e}"[:200]
        dist.broadcast_object_list(box, src=0)
        okc = torch.ones(1, dtype=torch.int64, device=dev)
        if box[0] is None:
            okc.zero_()
        else:
            # (in a thread with a deadline: a communicator that does not come up must not hang the bench - the ranks then agree on torch.distributed below)
            import threading
            res = {}
            def _make():
                try:
                    res["comm"] = ngsqc.Comm(rank, world, box[0], device=local_rank)
                except Exception as e:
                    res["err"] = e
            th = threading.Thread(target=_make, daemon=True); th.start(); th.join(float(os.environ.get("NGSQC_BENCH_COMM_TIMEOUT", "180")))
            if "comm" in res:
                comm = res["comm"]
            else:
                comm_note = (f"ngsqc_comm_init failed on rank {rank}: {res['err']}" if "err" in res else f"ngsqc_comm_init did not return on rank {rank} within its deadline")[:200]; okc.zero_()
        dist.all_reduce(okc, op=dist.ReduceOp.MIN)
        if int(okc.item()) == 0:
            if comm is not None:
                comm.close()
            comm = None; comm_note = comm_note or "a rank could not make the library's communicator: torch.distributed carries the collectives of this run"

    # ---- workload size: the full 30x file when the host can hold the image (plus a private copy per rank for N > 1) ----
    mode = 1 if args.ont else 0
    reads = args.reads
    size_note = ""
    if reads <= 0:
        if args.ont:
            reads = 400_000    # ~14 GB inflated: a shard of the 40x ONT file (8e6 reads); long-CIGAR stress, not a capacity test
            size_note = "shard of configs[4] (400 k of ~8e6 reads): the long-read generator is the limit, not the GPU"
        else:
            reads = FULL_READS
            avail = host_memory_available()
            need = reads * BYTES_PER_READ_COMPRESSED * (1.15 if world == 1 else 2.2)
            if avail and need > 0.8 * avail:
                reads = max(100_000_000, int(0.8 * avail / (BYTES_PER_READ_COMPRESSED * (1.15 if world == 1 else 2.2)) // 1_000_000 * 1_000_000))
                reads = min(reads, FULL_READS)
                size_note = f"scaled to {reads} reads: the host has {avail >> 30} GiB available for a {FULL_READS * BYTES_PER_READ_COMPRESSED >> 30} GiB image"

    # ---- synthetic batch (generated by host threads, outside the timed region) ----
    # N > 1: rank 0 generates ONE image with all cores, the others map it from /dev/shm (every rank still holds and processes its own copy
    # in HBM: same per-GPU work as one BAM per GPU; with --single-bam it is the one BAM that is sharded).
    t0 = time.time()
    image = None
    depth = 40.0 if args.ont else 30.0
    gen_kw = dict(seed=args.seed, mode=mode, depth=depth, first_contig=0, start_pos=0, level=args.level, aligned=True, flavor=0 if args.ont else args.flavor)
    share_name = f"ngsqc_bench_{os.environ.get('MASTER_PORT', '0')}_{args.seed}_{reads}.bam"
    share = None
    # SURVEY.md 8(d) config 4 is "seeds +0..7": one DIFFERENT BAM per GPU. When the host can hold world images at once every rank generates its own (seed + rank,
    # its share of the cores, all ranks at the same time); otherwise the ranks share one image through /dev/shm (same per-GPU work, identical counters per rank)
    per_rank_seed = False
    if world > 1 and not args.single_bam:
        box = [None]
        if rank == 0:
            avail = host_memory_available()
            # (N images are N times the generator's work: only when every rank gets at least 16 cores of its own - on a 16-CPU quota eight 30x images would take 40 minutes)
            cores_ok = G.effective_cpus() >= 16 * world or reads <= 50_000_000
            box[0] = bool(avail and avail > 1.3 * world * reads * BYTES_PER_READ_COMPRESSED and cores_ok and os.environ.get("NGSQC_BENCH_SHARED_IMAGE") is None)
        dist.broadcast_object_list(box, src=0)
        per_rank_seed = bool(box[0])
    if world > 1 and not per_rank_seed:
        ok = torch.zeros(1, dtype=torch.int64, device=dev)   # 1 + index of the directory that took the file, 0 = none
        if rank == 0:
            try:
                image = G.generate(reads, threads=0, **gen_kw)
            except MemoryError:
                image = None
            for k, d in enumerate(("/dev/shm", os.environ.get("TMPDIR", "/tmp"))):
                if image is None or not os.path.isdir(d):
                    continue
                try:
                    image.tofile(os.path.join(d, share_name)); ok += k + 1; break
                except OSError:
                    try:
                        os.remove(os.path.join(d, share_name))
                    except OSError:
                        pass
        dist.all_reduce(ok)
        if int(ok.item()) == 0:
            image = None   # (rank 0 too: every rank then generates the same kind of BAM below)
        if int(ok.item()) >= 1:
            share = os.path.join(("/dev/shm", os.environ.get("TMPDIR", "/tmp"))[int(ok.item()) - 1], share_name)
            if rank != 0:
                try:
                    image = np.memmap(share, dtype=np.uint8, mode="r")
                except OSError:
                    image = None
    if image is None and world > 1 and not per_rank_seed and not args.reads and reads > 96_000_000:
        # the shared image could not be written (no room in /dev/shm or TMPDIR): every rank must generate its own BAM with its share of the host's
        # cores - the full file would take world x 5 minutes, so the ranks fall back to a 96 M-read shard each and say so
        reads = 96_000_000
        size_note = "96 M-read shard per GPU: the generated 30x image could not be shared between the ranks on this host (no room in /dev/shm or TMPDIR)"
    if image is None and args.image_cache and os.path.exists(args.image_cache):
        image = np.fromfile(args.image_cache, dtype=np.uint8)
    if image is None:
        threads = max(1, (2 if world == 1 else 1) * G.effective_cpus() // max(world, 1))
        image = G.generate(reads, threads=threads, **dict(gen_kw, seed=args.seed + (0 if args.single_bam else rank)))
        if args.image_cache:
            image.tofile(args.image_cache)
    gen_s = time.time() - t0
    mark("image ready (generated / shared)")
    # H2D of the compressed image (of this rank's member range with --single-bam): not in the timed region, reported as end_to_end
    t0 = time.time()
    h = ngsqc.Handle(data=image, device=local_rank, shard=(rank, world)) if args.single_bam else ngsqc.Handle(data=image, device=local_rank)
    open_s = time.time() - t0
    mark("handle open (compressed image in HBM)")
    h2d_ms = h.timings()["h2d_ms"]
    if world > 1:
        dist.barrier()
        if rank == 0 and share:
            try:
                os.remove(share)
            except OSError:
                pass
    refs = h.refs
    omim = os.path.join(ROOT, "ngs-bits_amd", "resources", "hg38_440_omim_genes.bed")
    tx, ty = H.xy_tids(refs)
    nonspecial = H.nonspecial(refs)
    sites_arr = H.known_sites(refs)   # MappingQC's default third pass: pileup of the known common SNVs (Statistics::contamination); ~29 k sites for hg38

    # ---- the step of each tool ----
    tool = args.tool
    aux = {}
    if tool == "mappingqc":
        regs, _ = H.bed_regions(omim, refs, 3)
        mp = dict(mode=ngsqc.MODE_WGS, regions=regs, min_mapq=1, tid_x=tx, tid_y=ty, nonspecial=nonspecial)

        def job_step(hh, sharded):
            hh.drop_decoded()                                  # whole job from the compressed bytes, every step
            if sharded:
                # one decode per shard for the mapping scan AND the contamination pileup (ngsqc_run_job_partial); site counts are summed over the shards
                counters, _, _, _ = ngsqc.scan_mapping_sharded(hh, ngsqc.MODE_WGS, device=dev, comm=comm, regions=regs, min_mapq=1, tid_x=tx, tid_y=ty, nonspecial=nonspecial,
                                                               sites=sites_arr, site_params=(1, 13, args.ont))
            else:
                out = hh.run_job(mapping=mp, sites=sites_arr, site_params=(1, 13, args.ont))
                counters = out["counters"]
            roi_bases = int(counters[26]); usable_roi = int(counters[14])
            half = int(round(0.5 * usable_roi / roi_bases)) if roi_bases else 0
            hist, cov = hh.depth_stats(599, half)             # Histogram(0,599,5) input + half-depth count (Statistics.cpp:1185-1204)
            counters[27] = half; counters[28] = cov
            if world > 1 and not sharded:
                aux["local_counters"] = np.asarray(counters, dtype=np.int64).copy()
                # C1: RCCL reduce of the counter vectors over xGMI - the library's collective (ngsqc_comm_allreduce_counters); torch.distributed only with gloo
                counters = comm.allreduce_counters(counters) if comm is not None else ngsqc.allreduce_counters(counters, device=dev)
            return counters, hist

        def step():
            return job_step(h, args.single_bam)
    else:
        bed_path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"ngsqc_bench_exome_{args.seed}_{rank}.bed")
        n_lines = synthetic_exome_bed(bed_path, refs, args.seed, overlapping=(tool == "bedcoverage"))
        union, _ = H.bed_regions(bed_path, refs, 2)           # merge(true,true): the scanned regions
        lines, _ = H.bed_regions(bed_path, refs, 0)
        aux.update(bed_lines=n_lines, bed_bases=int(sum(e - s + 1 for _, s, e in union)))
        union_c = ngsqc.capi._regions_array(np.array(union, dtype=np.int32)); lines_c = ngsqc.capi._regions_array(np.array(lines, dtype=np.int32))   # C arrays built once
        n_union, n_lines_c = len(union), len(lines)

        def step():
            h.drop_decoded()
            if tool == "bedcoverage":
                h.scan_depth(union_c, min_mapq=1, n_regions=n_union)
                return h.region_sums(lines_c, n_lines=n_lines_c), None            # Statistics::avgCoverage: per-line depth sums
            h.scan_depth(union_c, min_mapq=1, min_baseq=args.min_baseq, n_regions=n_union)
            return h.lowhigh_runs(lines_c, 20, is_high=False, saturate254=True, n_lines=n_lines_c, as_array=True), None   # BedLowCoverage -cutoff 20 (sweep mode)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        result, hist = step()
        tms.append(h.timings())
        if os.environ.get("NGSQC_BENCH_VERBOSE"):
            print(f"[bench] step wall {1e3 * (time.perf_counter() - ts):.1f} ms, K1 {tms[-1]['inflate_ms']:.1f} ms", file=sys.stderr, flush=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    mark("warm-up + timed steps done")
    if world > 1:
        et = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
        elapsed = float(et.item())

    # the collective against a second channel: every rank's local counter vector gathered with torch.distributed and combined on the host must be what the
    # library's all-reduce returned on this rank
    coll = None
    if world > 1 and tool == "mappingqc" and not args.single_bam and "local_counters" in aux:
        loc = torch.tensor(aux["local_counters"], dtype=torch.int64, device=dev)
        parts = [torch.empty_like(loc) for _ in range(world)]
        dist.all_gather(parts, loc)
        expect = ngsqc.combine_counters_local([p_.cpu().numpy() for p_ in parts])
        oks = [None] * world
        dist.all_gather_object(oks, bool(np.array_equal(expect, np.asarray(result, dtype=np.int64))))
        distinct = len({tuple(p_.cpu().numpy()[:8].tolist()) for p_ in parts})
        coll = {"library": "libngsqc_hip ngsqc_comm_allreduce_counters (RCCL, loaded by the library)" if comm is not None else "torch.distributed (" + args.backend + ")",
                "allreduce_matches_gathered_sum_per_rank": oks, "distinct_inputs": distinct, "one_bam_per_gpu_with_its_own_seed": per_rank_seed}
        if comm_note:
            coll["note"] = comm_note
    n_rec = int(tms[-1]["n_records"])
    if args.single_bam:
        nr = torch.tensor([n_rec], dtype=torch.int64, device=dev)
        if world > 1:
            dist.all_reduce(nr)
        total_reads = int(nr.item()) * args.steps
    else:
        total_reads = n_rec * world * args.steps
    value = total_reads / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3

    def avg(k):
        return float(np.mean([t[k] for t in tms]))

    if rank == 0:
        c_bytes, u_bytes = int(tms[-1]["compressed_bytes"]), int(tms[-1]["inflated_bytes"])
        # ---- one extra un-pipelined step: K1 of a tile only starts when the previous tile is consumed, so the HIP-event stage times do
        # not overlap (t_scan = K2-K6 on resident inflated data) ----
        serial = None
        if world == 1:
            os.environ["NGSQC_PIPELINE"] = "0"
            ts = time.perf_counter(); step(); serial_wall = 1e3 * (time.perf_counter() - ts)
            serial = dict(h.timings(), wall_ms=serial_wall)
            del os.environ["NGSQC_PIPELINE"]
        infl_ms = avg("inflate_ms")
        huff_ms, lz_ms = avg("inflate_huff_ms"), avg("inflate_lz77_ms")
        k1_launches = max(1, int(tms[-1]["inflate_huff_launches"]))
        # ---- one more extra step with every K1 kernel in line on ONE stream (NGSQC_K1_SERIAL): a launch's HIP-event interval is then the kernel's own
        # duration (in the pipelined steps two phase-1 launches and phase 2 / CRC of earlier chunks share the chip, their intervals overlap) ----
        iso = None
        if world == 1:
            os.environ["NGSQC_PIPELINE"] = "0"; os.environ["NGSQC_K1_SERIAL"] = "1"
            step(); iso = h.timings()
            del os.environ["NGSQC_PIPELINE"]; del os.environ["NGSQC_K1_SERIAL"]
        src = iso if iso is not None else {"inflate_huff_ms": huff_ms, "inflate_lz77_ms": lz_ms}
        if src["inflate_huff_ms"] >= src["inflate_lz77_ms"]:
            dom = ("huff_tokens_kernel (K1 phase 1: Huffman decode, lane per BGZF member)", c_bytes / k1_launches, src["inflate_huff_ms"] / k1_launches)
        else:
            dom = ("lz77_groups_kernel (K1 phase 2: LZ77 window resolve, wave per BGZF member)", u_bytes / k1_launches, src["inflate_lz77_ms"] / k1_launches)
        dom_gbs = dom[1] / (dom[2] * 1e-3) / 1e9
        k1_gbs = (c_bytes + u_bytes) / (infl_ms * 1e-3) / 1e9
        scan_bytes = int(tms[-1]["scan_algorithmic_bytes"])
        shape = ("synthetic ONT-like BAM (configs[4] shape: N50 ~20 kb, 40x, ~1 CIGAR op per 12 bp, CG-tag records)" if args.ont else
                 f"synthetic 30x WGS BAM (configs[1] shape: 2x150 bp PE, coordinate-sorted, zlib-{args.level} BGZF" + ("" if not args.flavor else
                 f"; generator flavor {args.flavor}: " + ("SEQ from a synthetic reference genome, " if args.flavor & 1 else "random SEQ, ") + {0: "4-level", 1: "8-level", 2: "40-level"}[(args.flavor >> 1) & 3] + " QUAL") + ")")
        what = {"mappingqc": "MappingQC -wgs as one fused job (mapping_wgs + OMIM ROI depth + yxRatio + contamination pileup of the known SNVs; K6 depth histogram)",
                "bedcoverage": "BedCoverage (depth scan over the merged exome regions + per-line sums; unmerged BED with 10 % overlapping lines)",
                "bedlowcoverage": f"BedLowCoverage -cutoff 20{' -min_baseq ' + str(args.min_baseq) if args.min_baseq else ''} (depth scan + low-coverage runs, sweep saturation)"}[tool]
        full = (not args.ont) and n_rec >= FULL_READS
        out = {
            "metric": "Mreads/sec + achieved HBM GB/s, MappingQC 30x WGS BAM at 1/2/4/8 MI355X",
            "value": round(value, 3), "unit": "Mreads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if args.single_bam else "weak",
            "vs_baseline": None, "dtype": "u8/int32/int64", "data": "synthetic" if world == 1 else ("synthetic (one BAM per GPU, generated with seed + rank)" if per_rank_seed else "synthetic (one generated image, a private copy per rank in HBM)"),
            "config": {"workload": f"{what} on a {shape}; {'the full file' if full else 'reduced size'}: {n_rec} reads per GPU per step"
                                   f"{' (' + size_note + ')' if size_note else ''}; compressed image resident in HBM, streamed through {int(tms[-1]['n_tiles'])} tiles",
                       "reads_per_gpu_per_step": n_rec, "compressed_bytes_per_gpu": c_bytes, "inflated_bytes_per_gpu": u_bytes,
                       "tiles": int(tms[-1]["n_tiles"]), "k1_chunks": k1_launches, "members_inflated_per_step": int(tms[-1]["members_inflated"]), "bgzf_members": int(h.n_blocks),
                       "result_switches": (lambda sw: {"verify_crc": bool(sw & 1), "cram_ignore_md5": bool(sw & 2), "cram_no_reference": bool(sw & 4),
                                                       "at_defaults": sw == 1, "note": "ngsqc_timings.switches of the timed handle: the switches that can change a result"})(int(tms[-1].get("switches", 1))),
                       "roi": "hg38_440_omim_genes.bed (430 merged regions, 41.6 Mb)" if tool == "mappingqc" else f"synthetic exome BED ({aux.get('bed_lines')} lines, {aux.get('bed_bases')} merged bases)",
                       "parallelism": (f"one BAM sharded over {world} GPU(s) by BGZF member range, 1 process/GPU; all-gather of shard summaries, "
                                       "SUM all-reduce of counters and of the int32 difference array (RCCL)") if args.single_bam
                                      else f"{world} BAM(s), one per GPU, 1 process/GPU, RCCL all-reduce of the counter vectors"},
            "roofline": {"kernel": dom[0], "bound": "hbm", "limited_by": "instruction issue (VALU / LDS), not HBM: see profiles/r06_sq_counters.txt", "achieved": round(dom_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(dom_gbs / HBM_PEAK_GBS, 5), "traffic": None, "algorithmic_bytes_per_launch": int(dom[1]),
                         "avg_launch_ms": round(dom[2], 4), "launches_per_step": k1_launches, "sum_launches_ms": round(dom[2] * k1_launches, 3),
                         "isolated": iso is not None,
                         "pipelined_interval_ms": {"huff_tokens_kernel": round(huff_ms / k1_launches, 4), "lz77_groups_kernel": round(lz_ms / k1_launches, 4),
                                                   "note": "HIP-event intervals of the timed (pipelined) steps: co-resident launches on several streams, not kernel durations"},
                         "isolated_launch_ms": None if iso is None else {"huff_tokens_kernel": round(iso["inflate_huff_ms"] / k1_launches, 4), "lz77_groups_kernel": round(iso["inflate_lz77_ms"] / k1_launches, 4)},
                         "note": "avg_launch_ms: HIP events around the launch on its own stream in one extra step with the K1 kernels in line on one stream (the kernel alone on the "
                                 "chip), so launches x avg_launch_ms <= ms_per_step. achieved = algorithmic bytes of a launch (compressed bytes in for phase 1, inflated bytes out for "
                                 "phase 2) / that duration. DEFLATE decode is bit-serial per BGZF member: VALU-issue bound, not HBM bound (SURVEY.md §7)"},
            "roofline_k1_stage": {"kernels": "huff_tokens_kernel + lz77_groups_kernel + crc32_kernel (chunk stream on three HIP streams)", "achieved": round(k1_gbs, 2), "unit": "GB/s",
                                  "frac": round(k1_gbs / HBM_PEAK_GBS, 5), "algorithmic_bytes": c_bytes + u_bytes, "ms": round(infl_ms, 4)},
            "stage_ms": {"note": "sums of HIP-event intervals over the timed (pipelined) steps; index / scan / pileup overlap K1 of the next tile",
                         "inflate_huff": round(huff_ms, 4), "inflate_lz77": round(lz_ms, 4), "inflate_stage_wall": round(infl_ms, 4),
                         "index": round(avg("index_ms"), 4), "scan_kernels": round(avg("scan_kernel_ms"), 4), "scan_stage": round(avg("scan_ms"), 4),
                         "depth_finalize": round(avg("finalize_ms"), 4), "contamination_pileup": round(avg("pileup_ms"), 4), "job_wall": round(avg("job_wall_ms"), 4),
                         "contamination_sites": int(sites_arr.shape[0])},
            "end_to_end": {"h2d_ms": round(h2d_ms, 2), "h2d_GBps": round(c_bytes / max(h2d_ms, 1e-9) / 1e6, 2), "open_s": round(open_s, 2),
                           "value_incl_h2d": round(n_rec / (ms_per_step + h2d_ms) / 1e3, 3), "unit": "Mreads/s",
                           "sequential_value_incl_h2d": round(n_rec / (ms_per_step + h2d_ms) / 1e3, 3),
                           "sequential_note": "ngsqc_open_memory (the caller owns the buffer: the H2D completes inside open), then one resident step"},
            "host": {"cores_reported": os.cpu_count(), "cores_usable": G.effective_cpus(), "generate_s": round(gen_s, 2)},
        }
        if serial is not None:
            t_scan = serial["index_ms"] + serial["scan_ms"] + serial["finalize_ms"]   # K2-K6
            scan_gbs = scan_bytes / max(t_scan, 1e-9) / 1e6
            kern_gbs = scan_bytes / max(serial["scan_kernel_ms"], 1e-9) / 1e6
            out["roofline_scan"] = {"stage": "K2-K6 on resident inflated data (record index, scan kernels, depth prefix sum), un-pipelined step", "bound": "hbm",
                                    "achieved": round(scan_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(scan_gbs / HBM_PEAK_GBS, 5), "traffic": None,
                                    "algorithmic_bytes": scan_bytes, "t_scan_ms": round(t_scan, 4),
                                    "scan_kernel_only": {"achieved": round(kern_gbs, 2), "frac": round(kern_gbs / HBM_PEAK_GBS, 5), "ms": round(serial["scan_kernel_ms"], 4)}}
            out["stage_ms_unpipelined"] = {"inflate_stage": round(serial["inflate_ms"], 4), "inflate_huff": round(serial["inflate_huff_ms"], 4), "inflate_lz77": round(serial["inflate_lz77_ms"], 4),
                                           "index": round(serial["index_ms"], 4), "scan_stage": round(serial["scan_ms"], 4), "scan_kernels": round(serial["scan_kernel_ms"], 4),
                                           "depth_finalize": round(serial["finalize_ms"], 4), "contamination_pileup": round(serial["pileup_ms"], 4), "step_wall": round(serial["wall_ms"], 4)}
        # HBM traffic from the rocprofv3 PMC passes committed under profiles/ (separate --pmc runs of this command on a 48 M-read shard, K1 kernels in line):
        # raw FETCH_SIZE / WRITE_SIZE bytes per BGZF member (K1) or per record (scan stage), scaled to this run's launch
        try:
            per = {}
            settings_ok = False
            want = "k1_format=r06-word-per-trip-64B-lines tile_chunks=%s token_slots=%s" % (os.environ.get("NGSQC_TILE_CHUNKS", "8"), os.environ.get("NGSQC_TOKEN_SLOTS", "8"))
            for ln in open(os.path.join(ROOT, "profiles", "r06_hbm_traffic_pmc.txt")):
                if ln.startswith("# settings: "):
                    settings_ok = want in ln   # (the per-member figures only describe launches of the same kernels under the same schedule)
                if not settings_ok:
                    continue
                f = ln.rstrip("\n").split("\t")
                if len(f) >= 4 and f[1] in ("FETCH_SIZE", "WRITE_SIZE") and f[2] in ("bytes_per_member", "bytes_per_record"):
                    per[(f[0], f[1])] = float(f[3])
            members_per_launch = int(h.n_blocks) / k1_launches
            kname = dom[0].split(" ")[0]
            if (kname, "FETCH_SIZE") in per and (kname, "WRITE_SIZE") in per:
                fb, wb = per[(kname, "FETCH_SIZE")] * members_per_launch, per[(kname, "WRITE_SIZE")] * members_per_launch
                out["roofline"]["traffic"] = int(fb + wb)
                out["roofline"]["traffic_pmc"] = {"fetch_bytes_raw": int(fb), "write_bytes_raw": int(wb), "members_per_launch": int(members_per_launch),
                                                  "source": "profiles/r06_hbm_traffic_pmc.txt (raw counter bytes per BGZF member x the members of this run's launch)",
                                                  "ratio_to_algorithmic": round((fb + wb) / max(dom[1], 1), 2),
                                                  "note": "raw FETCH_SIZE / WRITE_SIZE x 1024 B (no x2: the K1 accesses are 16-byte pieces of 64 different member streams per "
                                                          "instruction, between the guide's narrow and wide regimes)"}
            if "roofline_scan" in out:
                keys = [k for k in per if k[0] in ("walk_scan_kernel", "scan_kernel", "index_count_kernel", "index_write_kernel", "index_guess_kernel")]
                if keys:
                    tot = sum(per[k] for k in keys) * n_rec
                    out["roofline_scan"]["traffic"] = int(tot)
                    out["roofline_scan"]["traffic_note"] = ("raw FETCH_SIZE + WRITE_SIZE of the K2 / scan kernels per record (profiles/r06_hbm_traffic_pmc.txt) x the records of the step; the "
                                                            "kernels gather one or two 128-byte lines per record, so the guide's x2 for wide streams does not apply")
        except OSError:
            pass
        if world == 1 and not args.no_cpu_baseline and tool == "mappingqc":
            import oracle_lib as O
            sample = min(args.cpu_sample_reads if not args.ont else 150_000, n_rec)
            c_cpu, st, secs = O.baseline_wgs_stream(image, omim, 1, sample, sites=sites_arr, site_params=(1, 13, bool(args.ont)))
            out["cpu_baseline"] = {"value": round(st["n_records"] / secs / 1e6, 4), "unit": "Mreads/s", "cores": 1, "kind": "port",
                                   "sample": f"first {st['n_records']} records of the same BAM ({st['compressed']} compressed bytes), oracle/stream.hpp single-thread sequential loop "
                                             f"(the GPU step's job: mapping_wgs + ROI depth + yxRatio + the contamination pileup of {int(sites_arr.shape[0])} known sites, all in one pass), {secs:.1f} s"}
            if args.ont:
                # long reads span BGZF members, so the file cannot be cut into per-thread member ranges: parity of the generator's data is checked on a
                # small BAM of the same generator instead (all 1032 counters of the GPU job vs the sequential oracle, bit-exact)
                small = G.generate(20_000, **dict(gen_kw, threads=0))
                hs = ngsqc.Handle(data=small, device=local_rank)
                got = hs.run_job(mapping=mp)["counters"]; hs.close()
                c_small, _, _ = O.baseline_wgs_stream(small, omim, 1, -1)
                skip = {27, 28}
                out["cpu_baseline"]["counters_match_gpu"] = bool(all(int(got[i]) == int(c_small[i]) for i in range(len(got)) if i not in skip))
                out["cpu_baseline"]["counters_match_note"] = "a 20 000-read BAM of the same generator and seed: all counters of the GPU job vs the sequential oracle, bit-exact"
            else:
                # the same loop on ALL host cores over the whole file: contiguous BGZF-member ranges per thread. Its additive counters and
                # the depth histogram are exact for an aligned BAM: the parity check of the bench input at full size
                try:
                    ncpu = G.effective_cpus()
                    st_mt, secs_mt, c_mt, hist_mt = O.baseline_wgs_stream_mt(image, omim, 1, 2 * ncpu, want_counters=True)
                    additive = np.ones(c_mt.size, dtype=bool); additive[list(O.ORDER_DEPENDENT)] = False
                    ok = bool(np.array_equal(c_mt[additive], np.asarray(result)[additive])) and bool(np.array_equal(hist_mt, hist)) and st_mt["n_records"] == n_rec
                    out["cpu_baseline_all_cores"] = {"value": round(st_mt["n_records"] / secs_mt / 1e6, 3), "unit": "Mreads/s", "cores": ncpu, "kind": "port",
                                                     "sample": f"the whole BAM ({st_mt['n_records']} records), {2 * ncpu} threads on {ncpu} usable CPUs (cgroup quota; the host reports "
                                                               f"{os.cpu_count()}), one contiguous BGZF-member range per thread, shared depth array, {secs_mt:.2f} s"}
                    out["cpu_baseline"]["counters_match_gpu"] = ok
                    out["cpu_baseline"]["counters_match_note"] = ("all-cores oracle over the WHOLE bench input vs the GPU's last timed step: every additive counter (1024 of 1032, incl. the "
                                                                  "insert-size histogram) and the 600-bin per-base depth histogram of the OMIM ROI, bit-exact; the order-dependent counters are "
                                                                  "covered by the parity tests")
                except Exception as e:   # never let the extra leg break the bench line
                    out["cpu_baseline_all_cores"] = {"error": str(e)[:200]}
        elif world == 1 and not args.no_cpu_baseline:
            # coverage tools: the oracle's restatement (1 thread) on the first records of the same BAM, and the same sample through the GPU path as parity check
            import oracle_lib as O
            samp = prefix_image(image, min(int(image.size), args.cpu_sample_reads * BYTES_PER_READ_COMPRESSED))
            sp = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"ngsqc_bench_sample_{args.seed}.bam")
            samp.tofile(sp)
            t1 = time.time()
            ob = O.Bam(sp)                                  # (sequential inflate + record framing of the sample: part of the CPU tool's work)
            hs = ngsqc.Handle(data=samp, device=local_rank)
            t_load = time.time() - t1; t1 = time.time()
            if tool == "bedcoverage":
                cov, _, _ = O.avg_coverage(ob, bed_path, merge_bed=False, min_mapq=1, random_access=False)
                secs = time.time() - t1 + t_load
                hs.scan_depth(union, min_mapq=1); ok = bool(np.array_equal(hs.region_sums(lines), cov))
            else:
                exp = O.low_high_coverage(ob, bed_path, 20, 1, args.min_baseq, is_high=False, random_access=False, tool_merge=1)
                secs = time.time() - t1 + t_load
                hs.scan_depth(union, min_mapq=1, min_baseq=args.min_baseq)
                ok = bool(np.array_equal(np.minimum(hs.depth(exp["roi_bases"]), 254), np.minimum(exp["depth"], 254)))
            hs.close(); os.remove(sp)
            out["cpu_baseline"] = {"value": round(ob.count / secs / 1e6, 4), "unit": "Mreads/s", "cores": 1, "kind": "port",
                                   "sample": f"first {ob.count} records of the same BAM: sequential inflate + record framing + the oracle's restatement of the tool's sweep, 1 thread, {secs:.1f} s", "counters_match_gpu": ok,
                                   "counters_match_note": "per-line depth sums (BedCoverage) / per-base depth of the whole exome BED (BedLowCoverage) of the sample, GPU vs oracle, bit-exact"}
    # ---- the other named configs of BASELINE.json in the SAME line: configs[2] = BedCoverage / BedLowCoverage over the resident 30x image,
    # configs[4] = the ONT shard (behind the headline handle: it needs the HBM) ----
    want_extra = rank == 0 and world == 1 and tool == "mappingqc" and not args.ont and not args.single_bam and os.environ.get("NGSQC_BENCH_NO_TOOLS") is None
    if want_extra:
        import oracle_lib as O
        out["tools"] = {}
        for tl, bq in (("bedcoverage", 0), ("bedlowcoverage", 0), ("bedlowcoverage", 20)):
            try:
                tag, leg = coverage_tool_leg(ngsqc, H, O, h, image, refs, tl, bq, max(3, min(args.steps, 5)), args, local_rank)
                out["tools"][tag] = leg
            except Exception as e:   # never let an extra leg break the bench line
                out["tools"][tl + (f"_baseq{bq}" if bq else "")] = {"error": str(e)[:300]}
    mark("stage / roofline / cpu-baseline / tool legs done")
    n_members_file = int(h.n_blocks) if not args.single_bam else None
    h.close()
    if want_extra and os.environ.get("NGSQC_BENCH_NO_ONT") is None:
        try:
            out["ont"] = ont_leg(ngsqc, G, H, O, args, local_rank, 3)
        except Exception as e:
            out["ont"] = {"error": str(e)[:300]}
    if want_extra and os.environ.get("NGSQC_BENCH_NO_FLAVORS") is None and not args.flavor:
        try:
            out["flavors"] = {"flavor5_refseq_40level_qual": flavor_leg(ngsqc, G, H, O, args, local_rank, 3)}
        except Exception as e:
            out["flavors"] = {"error": str(e)[:300]}
    if args.single_bam and rank == 0 and tool == "mappingqc" and image is not None:
        # parity of the sharded path: the unsharded job on this rank's GPU over the whole BAM (all 1032 counters and the depth histogram)
        try:
            hf = ngsqc.Handle(data=image, device=local_rank)
            of = hf.run_job(mapping=mp, sites=sites_arr, site_params=(1, 13, args.ont)); cf = of["counters"]
            rb, ur = int(cf[26]), int(cf[14]); half = int(round(0.5 * ur / rb)) if rb else 0
            hist_f, cov_f = hf.depth_stats(599, half); cf[27] = half; cf[28] = cov_f
            out["config"]["bgzf_members"] = int(hf.n_blocks)
            out["single_bam_parity"] = {"counters_match_one_gpu_job": bool(np.array_equal(cf, np.asarray(result))) and bool(np.array_equal(hist_f, hist)),
                                        "note": "all 1032 counters (order-dependent ones included) and the 600-bin depth histogram of the sharded step against the unsharded job over the whole BAM"}
            hf.close()
        except Exception as e:
            out["single_bam_parity"] = {"error": str(e)[:300]}

    # ---- the other way to use N GPUs, in the same line: ONE BAM sharded over the ranks by BGZF member range (configs[1] at N GPUs; `value` above is
    # configs[3], one BAM per GPU). Fused shard job per rank, all-gather of the summaries, SUM all-reduce of counters / site counts / difference array. ----
    strong = None
    if per_rank_seed and rank == 0:
        strong = {"skipped": "every rank holds a different BAM in this run (one BAM per GPU, seed + rank): the one-BAM-over-N-GPUs leg needs the same image on every rank - "
                             "run `bench.py --gpus N --single-bam`, or set NGSQC_BENCH_SHARED_IMAGE=1"}
    if tool == "mappingqc" and not args.ont and not args.single_bam and image is not None and not per_rank_seed and os.environ.get("NGSQC_BENCH_NO_STRONG") is None:
        try:
            hs = ngsqc.Handle(data=image, device=local_rank, shard=(rank, world))
            k2 = max(1, min(args.steps, 5))
            job_step(hs, True)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(); ts = time.perf_counter()
            for _ in range(k2):
                c_s, hist_s = job_step(hs, True)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(); el = time.perf_counter() - ts
            mem = torch.tensor([int(hs.timings()["members_inflated"]), int(hs.timings()["n_records"])], dtype=torch.int64, device=dev)
            if world > 1:
                et = torch.tensor([el], dtype=torch.float64, device=dev); dist.all_reduce(et, op=dist.ReduceOp.MAX); el = float(et.item())
                dist.all_reduce(mem)
            hs.close()
            if rank == 0:
                # (N > 1: `result` is the all-reduced vector of the N BAMs of the weak-scaling steps; the one-BAM job to compare with is rank 0's own, before the reduce)
                ref_c = np.asarray(aux["local_counters"] if world > 1 and "local_counters" in aux else result)
                same = bool(np.array_equal(np.asarray(c_s), ref_c)) and bool(np.array_equal(hist_s, hist))
                diff = [] if same else [(int(i), int(np.asarray(c_s)[i]), int(ref_c[i])) for i in np.nonzero(np.asarray(c_s) != ref_c)[0][:12]]
                strong = {"what": f"ONE 30x BAM over {world} GPU(s): shards by BGZF member range, ngsqc_run_job_partial per shard (mapping_wgs + contamination pileup in one decode), "
                                  "all-gather of the shard summaries, SUM all-reduce of counters, site counts and the int32 difference array",
                          "value": round(int(mem[1].item()) * k2 / el / 1e6, 3), "unit": "Mreads/s", "scaling": "strong", "steps": k2, "ms_per_step": round(el / k2 * 1e3, 3),
                          "members_inflated_per_step": int(mem[0].item()), "bgzf_members": n_members_file,
                          "counters_match_one_gpu_job": same, **({} if same else {"mismatch_index_sharded_unsharded": diff, "depth_histogram_matches": bool(np.array_equal(hist_s, hist))}),
                          "note": "all 1032 counters (the order-dependent ones included) and the 600-bin depth histogram against the unsharded job of rank 0; members behind a shard that "
                                  "only complete its last record are inflated by two shards (64 per cut)"}
        except Exception as e:   # never let the extra leg break the bench line
            if rank == 0:
                strong = {"error": str(e)[:300]}
    if rank == 0:
        if coll is not None:
            out["collective"] = coll
        if strong is not None:
            out["single_bam"] = strong
        # ---- end to end from a file: ngsqc_open(path) copies the image in the background while the first job already runs; and the tool itself ----
        if world == 1 and tool == "mappingqc" and not args.ont and os.environ.get("NGSQC_BENCH_NO_E2E") is None:
            try:
                import shutil
                # (a copy in /dev/shm is host memory: only when there is room for it next to the image itself)
                room = host_memory_available()
                d_ = next((d for d in ("/dev/shm", os.environ.get("TMPDIR", "/tmp")) if os.path.isdir(d) and shutil.disk_usage(d).free > image.size * 1.1
                           and (d != "/dev/shm" or not room or room > image.size * 1.5)), None)
                if d_ is not None:
                    bam_path = os.path.join(d_, f"ngsqc_bench_e2e_{args.seed}_{reads}.bam")
                    image.tofile(bam_path); open(bam_path + ".bai", "wb").close()   # (replaced by the index the library writes, below)
                    try:
                        te = time.perf_counter()
                        h2 = ngsqc.Handle(path=bam_path, device=local_rank)
                        t_open = time.perf_counter() - te
                        o2 = h2.run_job(mapping=mp, sites=sites_arr, site_params=(1, 13, args.ont))
                        t_e2e = time.perf_counter() - te
                        h2.upload_wait(); tm2 = h2.timings()
                        # the BAI index of the file (one more decode pass + the index kernels), then what SampleGender -method sry reads through it
                        try:
                            ti = time.perf_counter(); h2.write_bai(); t_bai = time.perf_counter() - ti
                            ti = time.perf_counter()
                            hp = ngsqc.Handle(path=bam_path, device=local_rank, regions=[("chrY", 2786989, 2787603)])
                            tid_y = [r[0] for r in hp.refs].index("chrY")
                            hp.scan_depth([(tid_y, 2786989, 2787603)], min_mapq=1); dsum = int(hp.depth(2787603 - 2786989 + 1).sum()); t_sry = time.perf_counter() - ti
                            tmp_ = hp.timings(); hp.close()
                            out["index"] = {"write_bai_s": round(t_bai, 3), "bai_bytes": os.path.getsize(bam_path + ".bai"),
                                            "region_query": "chrY:2786989-2787603 (the SRY window of SampleGender): ngsqc_open_regions + depth scan, wall clock",
                                            "region_query_s": round(t_sry, 4), "members_inflated": int(tmp_["members_inflated"]), "bgzf_members": n_members_file,
                                            "region_depth_sum": dsum}
                        except Exception as e:
                            out["index"] = {"error": str(e)[:300]}
                        h2.close()
                        out["end_to_end"].update({"open_plus_first_job_s": round(t_e2e, 3), "open_returns_after_s": round(t_open, 3), "h2d_ms": round(tm2["h2d_ms"], 2),
                                                  "h2d_GBps": round(c_bytes / max(tm2["h2d_ms"], 1e-9) / 1e6, 2), "value_incl_h2d": round(n_rec / t_e2e / 1e6, 3),
                                                  "counters_match": bool(np.array_equal(o2["counters"][:27], np.asarray(result)[:27])),
                                                  "note": "ngsqc_open(path of the page-cached file) + ONE job, wall clock: the compressed image is copied in the background (pieces in file "
                                                          "order, K1 chunks wait for their pieces), the tile-stream buffers are allocated beside it"})
                        tool_bin = os.path.join(ROOT, "ngs-bits_amd", "bin", "MappingQC")
                        if os.path.exists(tool_bin):
                            qc = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"ngsqc_bench_e2e_{args.seed}.qcML")
                            tt = time.perf_counter()
                            rc = subprocess.run([tool_bin, "-in", bam_path, "-wgs", "-build", "hg38", "-out", qc, "-no_ref"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600,
                                                env=dict(os.environ, NGSQC_TIMING="1"))
                            out["end_to_end"]["tool_wall_s"] = round(time.perf_counter() - tt, 3) if rc.returncode == 0 else None
                            # where the tool's wall time goes (seconds since process start, NGSQC_TIMING stamps of the host layer)
                            out["end_to_end"]["tool_stamps"] = [ln[len("[ngsqc] "):].strip() for ln in rc.stderr.decode(errors="replace").splitlines() if ln.startswith("[ngsqc] ")][:16]
                            out["end_to_end"]["tool"] = "bin/MappingQC -in <file> -wgs -build hg38 -out <qcML> -no_ref (process start to exit: open, fused job incl. contamination, qcML)"
                            if rc.returncode != 0:
                                out["end_to_end"]["tool_error"] = rc.stderr.decode(errors="replace")[-200:]
                            if os.path.exists(qc):
                                os.remove(qc)
                    finally:
                        for f_ in (bam_path, bam_path + ".bai"):
                            if os.path.exists(f_):
                                os.remove(f_)
            except Exception as e:
                out["end_to_end"]["e2e_error"] = str(e)[:300]
        mark("all legs done")
        out["budget"] = {"note": "wall time of this run by phase (seconds since the start of main on rank 0); the driver allows 1800 s", "phases": marks, "total_s": marks[-1][1]}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
