#!/bin/bash
# rocprofv3 passes behind profiles/r06_* (run on the GPU box from the repo root; each pass is its own run: PMC passes carry no trace domains).
# pass 1: kernel trace of the pipelined job. passes 2-6: with every K1 kernel in line on one stream and K1 of a tile not overlapped with the
# previous tile's consumers (NGSQC_K1_SERIAL=1 NGSQC_PIPELINE=0), so that a dispatch's counters are not mixed with a concurrent kernel's.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
export NGSQC_BENCH_NO_STRONG=1 NGSQC_BENCH_NO_E2E=1 NGSQC_BENCH_NO_TOOLS=1 NGSQC_BENCH_NO_ONT=1
CMD="python $R/bench.py --reads 48000000 --steps 3 --warmup 1 --no-cpu-baseline --image-cache /tmp/ngsqc_prof6_48m.bam"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/r6_trace -o t -- $CMD > $O/r6_trace.log 2>&1
export NGSQC_K1_SERIAL=1 NGSQC_PIPELINE=0
timeout 400 rocprofv3 --kernel-trace --stats -d $O/r6_trace_serial -o t -- $CMD > $O/r6_trace_serial.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/r6_fetch -o f -- $CMD > $O/r6_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/r6_write -o w -- $CMD > $O/r6_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY -d $O/r6_sq -o s -- $CMD > $O/r6_sq.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -d $O/r6_sq2 -o s -- $CMD > $O/r6_sq2.log 2>&1
cd $R
python tools/dev/save_profile.py r06 > $O/r6_save.log 2>&1
tail -2 $O/r6_trace.log $O/r6_fetch.log | cut -c1-300
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $O/r6_trace_ont -o t -- python $R/tools/dev/scan_probe.py --ont --reads 150000 --reps 2 --no-default --sets NGSQC_WALKERS=1 > $O/r6_trace_ont.log 2>&1)
python - <<PY > $O/r6_trace_ont.txt
import sqlite3,glob
for db in glob.glob("$O/r6_trace_ont/**/*_results.db", recursive=True):
    for r in sqlite3.connect(db).execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{r[0][:90]}\t{r[1]}\t{r[2]/1e3:.3f}\t{r[3]/1e3:.4f}\t{r[4]:.2f}")
PY
# the summaries travel back through gpurun_out/ (the box's own profiles/ directory does not); the raw rocprofv3 outputs stay on the box
cd $R; mkdir -p $O/profiles_r06; cp profiles/r06_* $O/profiles_r06/ 2>/dev/null; cp $O/r6_trace_ont.txt $O/profiles_r06/r06_kernel_stats_ont.txt 2>/dev/null
rm -rf $O/r6_trace $O/r6_trace_serial $O/r6_fetch $O/r6_write $O/r6_sq $O/r6_sq2 $O/r6_trace_ont
