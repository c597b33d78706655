#!/usr/bin/env python3
"""Dev: profiles/r02_bench_tools.txt from the bench.py lines of the other named workloads under gpurun_out/."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rows = [("bench_bedcoverage.json", "BedCoverage (configs[2]; unmerged exome BED, 10 % overlapping lines)"),
        ("bench_bedlowcoverage.json", "BedLowCoverage -cutoff 20"),
        ("bench_bedlowcoverageminbaseq20.json", "BedLowCoverage -cutoff 20 -min_baseq 20"),
        ("bench_ont.json", "MappingQC -wgs -single_end on ONT-like reads (configs[4])")]
out = ["# bench.py lines of the other named workloads (1x MI355X, round 2). Coverage tools: `python bench.py --tool <t> --reads 96000000 --steps 3 --warmup 1` - a 96 M-read shard",
       "# of the 30x BAM (the full file costs 5 minutes of BAM generation per run on the 16-CPU quota of the GPU box; throughput per read does not depend on the length of the",
       "# tile stream: K1 bound like MappingQC). ONT: `python bench.py --ont --steps 3 --warmup 1` (400 k reads = 14.4 GB inflated). Full JSON lines below the table.",
       "# workload\tMreads/s\tms/step\ttiles\tcpu_baseline (1 thread, Mreads/s)\tparity of the sample\troofline_scan.frac (K2-K6)\tK1 wall ms\tscan kernels ms"]
full = []
for f, name in rows:
    ln = open(os.path.join(ROOT, "gpurun_out", f)).read().strip().splitlines()[-1]
    d = json.loads(ln); cb = d.get("cpu_baseline", {})
    par = cb.get("counters_match_gpu", cb.get("parity"))
    out.append(f"{name}\t{d['value']}\t{d['ms_per_step']}\t{d['config'].get('tiles')}\t{cb.get('value')}\t{par}\t{d.get('roofline_scan', {}).get('frac')}\t"
               f"{d.get('stage_ms', {}).get('inflate_stage_wall')}\t{d.get('stage_ms', {}).get('scan_kernels')}")
    full += ["", f"## {name}", ln]
open(os.path.join(ROOT, "profiles", "r02_bench_tools.txt"), "w").write("\n".join(out + full) + "\n")
print("\n".join(out))
