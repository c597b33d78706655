#!/bin/bash
# dev: kernel timeline of the pipelined job on a probe of $1 reads (variant $2)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $O/k1_trace -o t --output-format csv -- python $R/tools/dev/job_probe.py ${1:-96000000} 1 ${2:-default} > $O/k1_trace.log 2>&1
python $R/tools/dev/timeline.py $O/k1_trace ${3:-110} > $O/k1_timeline.txt 2>&1
grep probe $O/k1_trace.log | cut -c1-330
cat $O/k1_timeline.txt
