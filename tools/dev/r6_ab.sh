#!/bin/bash
# dev (round 6): the full-size bench step with two builds of the library on ONE box (the generated image is kept between the runs): A the tree's, B a variant under ngs-bits_amd/libngsqc_hip_<tag>.so
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-ab}; mkdir -p $O; shift; TAGS="$@"
export NGSQC_BENCH_NO_STRONG=1 NGSQC_BENCH_NO_E2E=1 NGSQC_BENCH_NO_TOOLS=1 NGSQC_BENCH_NO_ONT=1 NGSQC_BENCH_NO_FLAVORS=1
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --image-cache /tmp/ngsqc_ab_full.bam"
$CMD > $O/a1.json 2> $O/a1.err
cp $R/ngs-bits_amd/libngsqc_hip.so /tmp/lib_a.so; RUNS="a1"
NGSQC_K1_CHUNK_WAVES=6 $CMD > $O/a1_cw6.json 2> $O/a1_cw6.err; RUNS="$RUNS a1_cw6"
for B in $TAGS; do cp $R/ngs-bits_amd/libngsqc_hip_$B.so $R/ngs-bits_amd/libngsqc_hip.so; $CMD > $O/$B.json 2> $O/$B.err; RUNS="$RUNS $B"; done
cp /tmp/lib_a.so $R/ngs-bits_amd/libngsqc_hip.so
$CMD > $O/a2.json 2> $O/a2.err; RUNS="$RUNS a2"
NGSQC_P1_STREAMS=3 $CMD > $O/a2_s3.json 2> $O/a2_s3.err; RUNS="$RUNS a2_s3"
python - <<PY
import json
for k in "$RUNS".split():
    try:
        d = json.loads(open("$O/" + k + ".json").read().strip().split("\n")[-1]); u = d["stage_ms_unpipelined"]; s = d["stage_ms"]
        print(k, d["value"], d["ms_per_step"], "unpipelined: inflate", u["inflate_stage"], "step", u["step_wall"], "scan", u["scan_stage"], "| pipelined: K1 wall", s["inflate_stage_wall"], "huff", s["inflate_huff"], "lz", s["inflate_lz77"], "scan", s["scan_kernels"], "index", s["index"])
    except Exception as e: print(k, "failed", e)
PY
rm -f /tmp/ngsqc_ab_full.bam
