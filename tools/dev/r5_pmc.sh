#!/bin/bash
# Round 5: counters of the chain walk alone (walk_scan_kernel; NGSQC_PIPELINE=0 steps of tools/dev/scan_probe.py on a 48 M-read shard, 1x MI355X).
# PMC passes carry no trace domains (gpurun refuses the combination); every pass is its own run of the same command.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
SETS=${1:-NGSQC_WALKERS=1}
CMD="python $R/tools/dev/scan_probe.py --reads 48000000 --reps 2 --no-default --image-cache /tmp/ngsqc_probe_48m.bam --sets $SETS"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/r5_counter_list.txt 2>&1 || rocprofv3 --list-avail > $O/r5_counter_list.txt 2>&1
pick() { for c in "$@"; do grep -q -w "$c" $O/r5_counter_list.txt && echo -n "$c "; done; }
G1=$(pick FETCH_SIZE)
G2=$(pick WRITE_SIZE)
G3=$(pick TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum)
G4=$(pick TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum)
G5=$(pick TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum)
G6=$(pick SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU)
G7=$(pick SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE)
i=0
for G in "$G1" "$G2" "$G3" "$G4" "$G5" "$G6" "$G7"; do
  i=$((i+1)); [ -z "$G" ] && continue
  timeout 300 rocprofv3 --pmc $G -d $O/r5_pmc$i -o p -- $CMD > $O/r5_pmc$i.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/r5_trace -o t -- $CMD > $O/r5_trace.log 2>&1
cd $R
python - <<'PY' > $O/r5_pmc_summary.txt 2>&1
import glob, sqlite3, os
for db in sorted(glob.glob("gpurun_out/r5_pmc*/**/*_results.db", recursive=True)):
    c = sqlite3.connect(db)
    try:
        for k, cn, n, s in c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
            if any(x in k for x in ("walk_scan", "index_", "scan_tile", "pileup", "depth_", "scan_long", "scan_kernel")):
                print(f"{k[:48]}\t{cn}\t{n}\t{s:.5e}\tper_dispatch={s/n:.5e}")
    except Exception as e:
        print(db, "error", e)
for db in sorted(glob.glob("gpurun_out/r5_trace/**/*_results.db", recursive=True)):
    c = sqlite3.connect(db)
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"TRACE\t{r[0][:60]}\t{r[1]}\t{r[2]/1e3:.3f}\t{r[3]/1e3:.4f}\t{r[4]:.2f}")
PY
tail -60 $O/r5_pmc_summary.txt
