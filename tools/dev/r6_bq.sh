#!/bin/bash
# dev (round 6): kernel trace of BedLowCoverage -min_baseq 20 on a probe shard
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-bq}; mkdir -p $O
export NGSQC_BENCH_NO_STRONG=1 NGSQC_BENCH_NO_E2E=1 NGSQC_BENCH_NO_TOOLS=1 NGSQC_BENCH_NO_ONT=1 NGSQC_BENCH_NO_FLAVORS=1
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --reads 48000000 --tool bedlowcoverage --min-baseq 20 --steps 3 --warmup 1 --no-cpu-baseline --image-cache /tmp/ngsqc_bq_48m.bam"
$CMD > $O/plain.json 2> $O/plain.err
NGSQC_PIPELINE=0 NGSQC_K1_SERIAL=1 timeout 400 rocprofv3 --kernel-trace --stats -d $O/t -o t --output-format csv -- $CMD > $O/trace.log 2>&1   # (every kernel alone on the chip)
NGSQC_BASEQ_RIDE=0 $CMD > $O/noride.json 2> $O/noride.err
python - <<PY
import json, glob, os
for f in ("plain", "noride"):
    try:
        d = json.loads(open("$O/" + f + ".json").read().strip().split("\n")[-1]); print(f, d["value"], d["ms_per_step"], d["roofline_scan"]["frac"], d["roofline_scan"]["t_scan_ms"], d["roofline_scan"].get("itemised_ms"))
    except Exception as e: print(f, "failed", e)
fs = sorted(glob.glob(os.path.join("$O", "t", "**", "*kernel_stats.csv"), recursive=True))
if fs:
    import csv
    for row in list(csv.reader(open(fs[-1])))[:26]: print(row[0][:70], *row[1:7])
PY
rm -rf $O/t
