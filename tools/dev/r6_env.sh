#!/bin/bash
# dev (round 6): the full-size bench step under several environments on ONE box (the generated image is kept between the runs). usage: r6_env.sh <tag> "K=V,K=V" "K=V" ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-env}; mkdir -p $O; shift
export NGSQC_BENCH_NO_STRONG=1 NGSQC_BENCH_NO_E2E=1 NGSQC_BENCH_NO_TOOLS=1 NGSQC_BENCH_NO_ONT=1 NGSQC_BENCH_NO_FLAVORS=1
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --image-cache /tmp/ngsqc_env_full.bam"
i=0
for spec in "" "$@"; do
  i=$((i+1)); ( for kv in $(echo "$spec" | tr ',' ' '); do export "$kv"; done; $CMD > $O/r$i.json 2> $O/r$i.err )
  python - <<PY
import json
try:
    d = json.loads(open("$O/r$i.json").read().strip().split("\n")[-1]); u = d["stage_ms_unpipelined"]; s = d["stage_ms"]
    print("[env] %-50s" % ("$spec" or "defaults"), d["value"], d["ms_per_step"], "unpipelined: inflate", u["inflate_stage"], "step", u["step_wall"], "| pipelined: K1 wall", s["inflate_stage_wall"], "huff", s["inflate_huff"], "lz", s["inflate_lz77"], "scan", s["scan_kernels"], "index", s["index"], "tiles", d["config"]["tiles"], "chunks", d["config"]["k1_chunks"])
except Exception as e: print("[env] $spec failed", e)
PY
done
rm -f /tmp/ngsqc_env_full.bam
