#!/bin/bash
# dev (round 6): the decoder's LDS-DMA input path - tests, then the probe with counted waits against vmcnt(0) waits, pipelined and with every kernel alone
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-dma}; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_parity.py tests/test_gpu_lowhigh.py tests/test_gpu_synthetic.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/tools/dev/job_probe.py ${2:-96000000} 5 default,dmasafe,serial,serial_dmasafe > $O/probe.log 2>&1
grep probe $O/probe.log | cut -c1-400
