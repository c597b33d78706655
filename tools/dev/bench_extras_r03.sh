#!/bin/bash
# round 3: bench lines of the other named workloads on a 96 M-read shard (+ the realistic-entropy generator flavors, + a token-ring variant of the job)
O=gpurun_out/r3_extras; mkdir -p $O
export NGSQC_BENCH_NO_STRONG=1 NGSQC_BENCH_NO_E2E=1
C=/dev/shm/ngsqc_extras_f0.bam
python bench.py --reads 96000000 --steps 3 --warmup 1 --tool bedcoverage --image-cache $C > $O/bedcoverage.json 2> $O/bedcoverage.err
python bench.py --reads 96000000 --steps 3 --warmup 1 --tool bedlowcoverage --image-cache $C > $O/bedlowcoverage.json 2> $O/bedlowcoverage.err
python bench.py --reads 96000000 --steps 3 --warmup 1 --tool bedlowcoverage --min-baseq 20 --image-cache $C > $O/bedlowcoverage_bq20.json 2> $O/bedlowcoverage_bq20.err
rm -f $C
python bench.py --reads 96000000 --steps 3 --warmup 1 --flavor 3 > $O/mappingqc_flavor3.json 2> $O/mappingqc_flavor3.err
python bench.py --reads 96000000 --steps 3 --warmup 1 --flavor 5 > $O/mappingqc_flavor5.json 2> $O/mappingqc_flavor5.err
python bench.py --ont --steps 3 --warmup 1 > $O/ont.json 2> $O/ont.err
python tools/dev/job_probe.py 96000000 5 default,slots3,default,slots3 > $O/job_probe.log 2>&1
for f in $O/*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("counters_match_gpu", d.get("cpu_baseline", {}).get("parity")), d["config"].get("compressed_bytes_per_gpu"), d["roofline"].get("isolated_launch_ms"))
except Exception as e:
    print("ERR", e)
PY
done
tail -8 $O/job_probe.log
