#!/bin/bash
# round 4: the r04 profile set again (the walk / pileup kernels changed after the first take) and bench lines of the other named workloads on 96 M-read shards
O=gpurun_out/r4_extras; mkdir -p $O
tools/dev/profile_r04.sh > $O/profile.log 2>&1; cp profiles/r04_* $O/ 2>/dev/null
export NGSQC_BENCH_NO_STRONG=1 NGSQC_BENCH_NO_E2E=1
C=/dev/shm/ngsqc_extras_f0.bam
timeout 200 python bench.py --reads 96000000 --steps 3 --warmup 1 --tool bedcoverage --image-cache $C > $O/bedcoverage.json 2> $O/bedcoverage.err
timeout 200 python bench.py --reads 96000000 --steps 3 --warmup 1 --tool bedlowcoverage --image-cache $C > $O/bedlowcoverage.json 2> $O/bedlowcoverage.err
rm -f $C
timeout 200 python bench.py --reads 96000000 --steps 3 --warmup 1 --flavor 3 --no-cpu-baseline > $O/mappingqc_flavor3.json 2> $O/mappingqc_flavor3.err
timeout 300 python bench.py --ont --steps 3 --warmup 1 > $O/ont.json 2> $O/ont.err
for f in $O/*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("counters_match_gpu", d.get("cpu_baseline", {}).get("parity")), d["config"].get("compressed_bytes_per_gpu"), d["roofline"].get("isolated_launch_ms"), d.get("roofline_scan", {}).get("frac"))
except Exception as e:
    print("ERR", e)
PY
done
