#!/bin/bash
# dev (round 6): K1 after a kernel change - the inflate tests, the fused job on a probe shard (pipelined and with every kernel alone), the instruction
# counters and the kernel trace of the K1 kernels alone. usage: r6_k1.sh [tag] [reads] [tests: 0/1]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-k1}; N=${2:-48000000}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
if [ "${3:-1}" = "1" ]; then timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_parity.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log; fi
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/tools/dev/job_probe.py $N 5 default,serial > $O/probe.log 2>&1
grep probe $O/probe.log | cut -c1-400
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY -d $O/a -o s --output-format csv -- python $R/tools/dev/job_probe.py $N 1 serial > $O/a.log 2>&1
if [ "${R6_WRITE:-0}" = "1" ]; then
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/w -o s --output-format csv -- python $R/tools/dev/job_probe.py $N 1 serial > $O/w.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/f -o s --output-format csv -- python $R/tools/dev/job_probe.py $N 1 serial > $O/f.log 2>&1
fi
timeout 300 rocprofv3 --kernel-trace --stats -d $O/c -o s --output-format csv -- python $R/tools/dev/job_probe.py $N 1 serial > $O/c.log 2>&1
python - <<PY > $O/summary.txt 2>&1
import csv, glob, os, collections
fs = sorted(glob.glob(os.path.join("$O", "a", "**", "*counter_collection.csv"), recursive=True))
if fs:
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[-1])):
        k = r['Kernel_Name'].split('(')[0].split('::')[-1][:24]
        acc[(k, r['Counter_Name'])] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    for (k, c) in sorted(acc):
        if any(x in k for x in ('huff', 'lz77', 'crc32')): print(f"{k:26s} {c:22s} {cnt[(k,c)]:5d} {acc[(k,c)]:.4e}")
for sub in ("w", "f"):
    fs = sorted(glob.glob(os.path.join("$O", sub, "**", "*counter_collection.csv"), recursive=True))
    if fs:
        acc = collections.defaultdict(float)
        for r in csv.DictReader(open(fs[-1])):
            k = r['Kernel_Name'].split('(')[0].split('::')[-1][:24]
            acc[(k, r['Counter_Name'])] += float(r['Counter_Value'])
        for (k, c) in sorted(acc):
            if any(x in k for x in ('huff', 'lz77', 'crc32')): print(f"{k:26s} {c:22s} {acc[(k,c)]:.4e}")
fs = sorted(glob.glob(os.path.join("$O", "c", "**", "*kernel_stats.csv"), recursive=True))
if fs: print(open(fs[-1]).read()[:2500])
PY
cat $O/summary.txt
rm -rf $O/a $O/c $O/w $O/f
