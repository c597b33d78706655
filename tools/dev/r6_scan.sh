#!/bin/bash
# dev (round 6): the scan stage after a change of the walk - the whole GPU suite, then the un-pipelined scan stage of MappingQC (short reads) and of the ONT shape
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-scan}; mkdir -p $O
cd $R
if [ "${2:-1}" = "1" ]; then timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log; fi
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/tools/dev/scan_probe.py --reads 96000000 --reps 3 --sets "" > $O/short.log 2>&1; grep -v "^\[probe\] gen" $O/short.log | tail -3 | cut -c1-300
timeout 900 python $R/tools/dev/scan_probe.py --ont --reads 400000 --reps 3 --sets "" > $O/ont.log 2>&1; tail -3 $O/ont.log | cut -c1-300
