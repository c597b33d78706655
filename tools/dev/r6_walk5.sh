#!/bin/bash
# dev (round 6): the coverage tools' legs under environments on a shard of whole tiles. usage: r6_walk5.sh <outdir> <reads> <spec> ...   spec = ENV=V,ENV=V (or "-" for the defaults)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-walk5}; mkdir -p $O; N=${2:-189000000}; shift; shift
export NGSQC_BENCH_NO_STRONG=1 NGSQC_BENCH_NO_E2E=1 NGSQC_BENCH_NO_TOOLS=1 NGSQC_BENCH_NO_ONT=1 NGSQC_BENCH_NO_FLAVORS=1
cd /tmp && export TMPDIR=/tmp
i=0
for T in "bedcoverage" "bedlowcoverage --min-baseq 20"; do
  for S in "$@"; do
    i=$((i+1)); E=""; [ "$S" != "-" ] && E=$(echo $S | tr ',' ' ')
    env $E python $R/bench.py --reads $N --tool $T --steps 4 --warmup 1 --no-cpu-baseline --image-cache /tmp/ngsqc_w5.bam > $O/r$i.json 2> $O/r$i.err
    python - <<PY
import json
try:
    d = json.loads(open("$O/r$i.json").read().strip().split("\n")[-1]); print("$T | $S |", d["value"], d["ms_per_step"], d["roofline_scan"]["frac"], d["roofline_scan"]["t_scan_ms"], d["roofline_scan"].get("itemised_ms"), d["config"].get("tiles"), flush=True)
except Exception as e: print("$T | $S | failed", e, flush=True)
PY
  done
done
rm -f /tmp/ngsqc_w5.bam
