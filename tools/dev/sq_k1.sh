#!/bin/bash
# dev: SQ counters of the K1 kernels alone (every kernel in line on one stream) on a probe shard; usage: sq_k1.sh [reads] [extra switches, e.g. K1_CHUNK_MUL=2+P1_WAVES=12]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
V=serial; [ -n "${2:-}" ] && V="serial+$2"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY -d $O/k1_sq -o s --output-format csv -- python $R/tools/dev/job_probe.py ${1:-48000000} 1 $V > $O/k1_sq.log 2>&1
python - <<'PY' > $O/k1_sq.txt 2>&1
import csv, glob, os, collections
f = sorted(glob.glob(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out/k1_sq', '**', '*counter_collection.csv'), recursive=True))[-1]
acc = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].split('::')[-1][:24]
    acc[(k, r['Counter_Name'])] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for (k, c) in sorted(acc):
    if any(x in k for x in ('huff', 'lz77', 'crc32')): print(f"{k:26s} {c:22s} {cnt[(k,c)]:5d} {acc[(k,c)]:.4e}")
PY
grep probe $O/k1_sq.log | cut -c1-330
cat $O/k1_sq.txt
