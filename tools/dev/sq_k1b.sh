#!/bin/bash
# dev: instruction and stall counters of the K1 kernels alone (two PMC passes) on a probe shard; usage: sq_k1b.sh [reads] [tag]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${2:-sq}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/tools/dev/job_probe.py ${1:-48000000} 1 serial > $O/warm.log 2>&1   # (generates and caches the shard)
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY -d $O/a -o s --output-format csv -- python $R/tools/dev/job_probe.py ${1:-48000000} 1 serial > $O/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -d $O/b -o s --output-format csv -- python $R/tools/dev/job_probe.py ${1:-48000000} 1 serial > $O/b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/c -o s --output-format csv -- python $R/tools/dev/job_probe.py ${1:-48000000} 1 serial > $O/c.log 2>&1
python - <<PY > $O/summary.txt 2>&1
import csv, glob, os, collections
for sub in ("a", "b"):
    fs = sorted(glob.glob(os.path.join("$O", sub, "**", "*counter_collection.csv"), recursive=True))
    if not fs: print(sub, "no counter file"); continue
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[-1])):
        k = r['Kernel_Name'].split('(')[0].split('::')[-1][:24]
        acc[(k, r['Counter_Name'])] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    for (k, c) in sorted(acc):
        if any(x in k for x in ('huff', 'lz77', 'crc32', 'walk_scan')): print(f"{k:26s} {c:22s} {cnt[(k,c)]:5d} {acc[(k,c)]:.4e}")
fs = sorted(glob.glob(os.path.join("$O", "c", "**", "*kernel_stats.csv"), recursive=True))
if fs: print(open(fs[-1]).read()[:3000])
PY
grep probe $O/a.log | cut -c1-330
cat $O/summary.txt
