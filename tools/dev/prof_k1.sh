#!/bin/bash
# dev: kernel timeline of the pipelined job + SQ counters of the K1 kernels alone (serial), on a 96M-read probe
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/k1_trace -o t --output-format csv -- python $R/tools/dev/job_probe.py ${1:-96000000} 2 default > $O/k1_trace.log 2>&1
python $R/tools/dev/timeline.py $O/k1_trace 70 > $O/k1_timeline.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY -d $O/k1_sq -o s --output-format csv -- python $R/tools/dev/job_probe.py ${1:-96000000} 1 serial > $O/k1_sq.log 2>&1
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
grep probe $O/k1_trace.log $O/k1_sq.log | cut -c1-330
cat $O/k1_sq.txt
tail -70 $O/k1_timeline.txt
