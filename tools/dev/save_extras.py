"""dev: profiles/r03_bench_tools.txt from the JSON lines of tools/dev/bench_extras_r03.sh (gpurun_out/r3_extras/*.json)"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(R, "gpurun_out", "r3_extras")
rows = [("bedcoverage", "BedCoverage (configs[2]; unmerged exome BED, 10 % overlapping lines)", "python bench.py --reads 96000000 --steps 3 --warmup 1 --tool bedcoverage"),
        ("bedlowcoverage", "BedLowCoverage -cutoff 20", "… --tool bedlowcoverage"),
        ("bedlowcoverage_bq20", "BedLowCoverage -cutoff 20 -min_baseq 20", "… --tool bedlowcoverage --min-baseq 20"),
        ("mappingqc_flavor3", "MappingQC -wgs, generator flavor 3 (SEQ from a synthetic reference, 8-level QUAL: NovaSeq-like entropy)", "… --flavor 3"),
        ("mappingqc_flavor5", "MappingQC -wgs, generator flavor 5 (SEQ from a synthetic reference, 40-level QUAL)", "… --flavor 5"),
        ("ont", "MappingQC -wgs -single_end on ONT-like reads (configs[4])", "python bench.py --ont --steps 3 --warmup 1")]
out = ["# bench.py lines of the other named workloads (1x MI355X, round 3; NGSQC_BENCH_NO_STRONG=1 NGSQC_BENCH_NO_E2E=1: the main line only). Short reads: a 96 M-read shard of the 30x BAM",
       "# (the full file costs 5 minutes of BAM generation per run on the 16-CPU quota of the GPU box). ONT: 400 k reads. Full JSON lines below the table.",
       "# workload\tMreads/s\tms/step\ttiles\tcompressed GB\tinflated GB\tcpu_baseline (1 thread, Mreads/s)\tparity of the sample\troofline_scan.frac (K2-K6)\tK1 wall ms\thuff / lz77 alone, ms per launch"]
full = []
for key, title, cmd in rows:
    p = os.path.join(src, key + ".json")
    if not os.path.exists(p):
        continue
    line = open(p).read().strip().splitlines()[-1]; d = json.loads(line)
    iso = d["roofline"].get("isolated_launch_ms", {})
    out.append("\t".join(str(x) for x in (title, d["value"], d["ms_per_step"], d["config"].get("tiles"), round(d["config"]["compressed_bytes_per_gpu"] / 1e9, 2), round(d["config"]["inflated_bytes_per_gpu"] / 1e9, 2),
                                         d["cpu_baseline"]["value"], d["cpu_baseline"].get("counters_match_gpu"), d["roofline_scan"]["frac"], d["stage_ms"].get("inflate_stage_wall"),
                                         f'{iso.get("huff_tokens_kernel")} / {iso.get("lz77_groups_kernel")}')))
    full += ["", f"## {title}   ({cmd})", line]
open(os.path.join(R, "profiles", "r03_bench_tools.txt"), "w").write("\n".join(out + full) + "\n")
print("\n".join(out))
