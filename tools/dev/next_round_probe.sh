#!/bin/bash
# First GPU call of the next round (about 6 GPU-minutes): the decoder kernel at 3 waves per SIMD (166 VGPRs, no scratch) against the shipped 4 (128 VGPRs,
# ScratchSize 156: ~20 scratch stores / reloads per group of four trips, DESIGN.md section 7 "What comes next for K1" (5)).
#   here, before the call :  bash tools/dev/next_round_probe.sh build     (cross-compiles the variant into ngs-bits_amd/variants/libngsqc_hip_p1w3.so; git-ignored, travels with the snapshot)
#   on the box            :  bash tools/dev/next_round_probe.sh run       (same image, shipped library and variant alternating; value and the K1 stage times of every run)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
if [ "${1:-run}" = build ]; then
	T=$(mktemp -d) && cp -r $R/ngs-bits_amd/csrc $T/csrc && mkdir -p $T/include $R/ngs-bits_amd/variants && cp $R/include/ngsqc.h $T/include/
	sed -i 's/constexpr int P1_WAVES_PER_SIMD = 4;/constexpr int P1_WAVES_PER_SIMD = 3;/' $T/csrc/k1_kernels.h
	sed -i 's#"../../include/ngsqc.h"#"'$T'/include/ngsqc.h"#' $T/csrc/common.h
	(cd $T/csrc && rm -f *.o && make -s -j8 LIB=$R/ngs-bits_amd/variants/libngsqc_hip_p1w3.so) && ls -la $R/ngs-bits_amd/variants/
	exit 0
fi
O=$R/gpurun_out/r5_probe; mkdir -p $O
export NGSQC_BENCH_NO_STRONG=1 NGSQC_BENCH_NO_E2E=1
CMD="python $R/bench.py --reads 96000000 --steps 5 --warmup 1 --no-cpu-baseline --image-cache /tmp/ngsqc_probe_96m.bam"
cp $R/ngs-bits_amd/libngsqc_hip.so /tmp/libngsqc_hip_shipped.so
for v in shipped p1w3 shipped p1w3; do
	if [ $v = shipped ]; then cp /tmp/libngsqc_hip_shipped.so $R/ngs-bits_amd/libngsqc_hip.so; else cp $R/ngs-bits_amd/variants/libngsqc_hip_$v.so $R/ngs-bits_amd/libngsqc_hip.so; fi
	timeout 400 $CMD > $O/$v.$RANDOM.json 2> $O/$v.err; f=$(ls -t $O/$v.*.json | head -1)
	python - "$v" "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    st = d.get("stages", {}) or {}
    print(sys.argv[1], "value", d.get("value"), "ms_per_step", d.get("ms_per_step"), {k: st[k] for k in st if "inflate" in k or "k1" in k.lower()})
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
done
cp /tmp/libngsqc_hip_shipped.so $R/ngs-bits_amd/libngsqc_hip.so
