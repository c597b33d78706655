#!/bin/bash
# dev (round 6): schedule variants of the K1 chunk stream on a probe shard long enough for a steady state. usage: r6_sched.sh [tag] [reads] [variants]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-sched}; N=${2:-192000000}; V=${3:-default}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 1500 python $R/tools/dev/job_probe.py $N 4 $V > $O/probe.log 2>&1
grep probe $O/probe.log | cut -c1-260
