#!/bin/bash
# dev (round 6): kernel trace of the BedCoverage leg (un-pipelined, every kernel alone on the chip)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-tooltrace}; mkdir -p $O
export NGSQC_BENCH_NO_STRONG=1 NGSQC_BENCH_NO_E2E=1 NGSQC_BENCH_NO_TOOLS=1 NGSQC_BENCH_NO_ONT=1 NGSQC_BENCH_NO_FLAVORS=1
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --reads 126000000 --tool bedcoverage --steps 3 --warmup 1 --no-cpu-baseline --image-cache /tmp/ngsqc_tt_126m.bam"
$CMD > $O/plain.json 2> $O/plain.err
NGSQC_PIPELINE=0 NGSQC_K1_SERIAL=1 timeout 400 rocprofv3 --kernel-trace --stats -d $O/t -o t --output-format csv -- $CMD > $O/trace.log 2>&1
python - <<PY
import json, glob, os, csv
d = json.loads(open("$O/plain.json").read().strip().split("\n")[-1]); print(d["value"], d["ms_per_step"], d["roofline_scan"]["frac"], d["roofline_scan"]["t_scan_ms"], d["roofline_scan"].get("itemised_ms"), d["config"].get("tiles"))
fs = sorted(glob.glob(os.path.join("$O", "t", "**", "*kernel_stats.csv"), recursive=True))
if fs:
    for row in list(csv.reader(open(fs[-1])))[:40]: print(row[0][:80], *row[1:7])
PY
rm -rf $O/t /tmp/ngsqc_tt_126m.bam
