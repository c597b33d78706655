#!/bin/bash
# dev: wall time of bin/MappingQC -wgs on a generated BAM in /dev/shm, with the stamps of NGSQC_TIMING
R=${GRAFT_REPO_ROOT:-$(pwd)}; N=${1:-96000000}
F=/dev/shm/ngsqc_tool_$N.bam
python - <<PY
import sys; sys.path.insert(0, "$R/tests")
import bamgen_lib as G
G.generate($N).tofile("$F"); open("$F.bai", "wb").close()
PY
for i in 1 2; do T0=$(date +%s.%N); NGSQC_TIMING=1 $R/ngs-bits_amd/bin/MappingQC -in $F -wgs -build hg38 -out /tmp/tool_probe.qcML -no_ref 2>&1 | grep "ngsqc" | cut -c1-200; T1=$(date +%s.%N); echo "[tool] wall $(echo "$T1 - $T0" | bc) s"; done
rm -f $F $F.bai /tmp/tool_probe.qcML
