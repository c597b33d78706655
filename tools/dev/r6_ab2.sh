#!/bin/bash
# dev (round 6): the full-size bench step under builds of the library and environments on ONE box (the generated image is kept between the runs).
# usage: r6_ab2.sh <outdir> <spec> ...   spec = tag[:ENV=V,ENV=V]; tag "a" = the tree's build, any other tag = ngs-bits_amd/libngsqc_hip_<tag>.so
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-ab2}; mkdir -p $O; shift
export NGSQC_BENCH_NO_STRONG=1 NGSQC_BENCH_NO_E2E=1 NGSQC_BENCH_NO_TOOLS=1 NGSQC_BENCH_NO_ONT=1 NGSQC_BENCH_NO_FLAVORS=1
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --image-cache /tmp/ngsqc_ab_full.bam"
cp $R/ngs-bits_amd/libngsqc_hip.so /tmp/lib_a.so
i=0
for S in "$@"; do
  i=$((i+1)); T=${S%%:*}; E=""; [ "$S" != "$T" ] && E=${S#*:}
  if [ "$T" = "a" ]; then cp /tmp/lib_a.so $R/ngs-bits_amd/libngsqc_hip.so; else cp $R/ngs-bits_amd/libngsqc_hip_$T.so $R/ngs-bits_amd/libngsqc_hip.so; fi
  env $(echo $E | tr ',' ' ') $CMD > $O/r$i.json 2> $O/r$i.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/r$i.json").read().strip().split("\n")[-1]); u = d["stage_ms_unpipelined"]; s = d["stage_ms"]
    print("$S", d["value"], d["ms_per_step"], "unpipelined: inflate", u["inflate_stage"], "step", u["step_wall"], "scan", u["scan_stage"], "| pipelined: K1 wall", s["inflate_stage_wall"], "huff", s["inflate_huff"], "lz", s["inflate_lz77"], "scan", s["scan_stage"], "index", s["index"], "| isolated", d["roofline"]["isolated_launch_ms"], flush=True)
except Exception as e: print("$S", "failed", e, flush=True)
PY
done
cp /tmp/lib_a.so $R/ngs-bits_amd/libngsqc_hip.so
rm -f /tmp/ngsqc_ab_full.bam
