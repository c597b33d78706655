#!/bin/bash
# dev (round 6): the coverage tools' walk compiled for 3 / 5 waves per SIMD on three whole tiles of the 30x shape
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-walk5}; mkdir -p $O
export NGSQC_BENCH_NO_STRONG=1 NGSQC_BENCH_NO_E2E=1 NGSQC_BENCH_NO_TOOLS=1 NGSQC_BENCH_NO_ONT=1 NGSQC_BENCH_NO_FLAVORS=1
cd /tmp && export TMPDIR=/tmp
for T in "bedcoverage" "bedlowcoverage --min-baseq 20"; do
  for W in 3 5 3 5; do
    N=$(echo $T | tr -d ' -')_w$W
    NGSQC_WALK_WAVES=$W python $R/bench.py --reads 189000000 --tool $T --steps 4 --warmup 1 --no-cpu-baseline --image-cache /tmp/ngsqc_w5_189m.bam > $O/$N.json 2> $O/$N.err
    python - <<PY
import json
try:
    d = json.loads(open("$O/$N.json").read().strip().split("\n")[-1]); print("$N", d["value"], d["ms_per_step"], d["roofline_scan"]["frac"], d["roofline_scan"]["t_scan_ms"], d["roofline_scan"].get("itemised_ms"), d["config"].get("tiles"))
except Exception as e: print("$N", "failed", e)
PY
  done
done
rm -f /tmp/ngsqc_w5_189m.bam
