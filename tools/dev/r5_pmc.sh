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
# (a pass takes at most a few TCC / TCP counters: more "exceeds the capabilities of the hardware to collect")
G1=$(pick TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum)
G2=$(pick TCC_HIT_sum TCC_MISS_sum)
G3=$(pick TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum)
G4=${R5_PMC_MORE:+$(pick TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum)}
G5=""; G6=""; G7=""
i=0
for G in "$G1" "$G2" "$G3" "$G4" "$G5" "$G6" "$G7"; do
  i=$((i+1)); [ -z "$G" ] && continue
  timeout 300 rocprofv3 --pmc $G -d $O/r5b_pmc$i -o p -- $CMD > $O/r5b_pmc$i.log 2>&1
done
[ -n "${R5_PMC_TRACE:-}" ] && timeout 300 rocprofv3 --kernel-trace --stats -d $O/r5_trace -o t -- $CMD > $O/r5_trace.log 2>&1
cd $R
python - <<'PY' > $O/r5b_pmc_summary.txt 2>&1
import glob, sqlite3, os
for db in sorted(glob.glob("gpurun_out/r5b_pmc*/**/*_results.db", recursive=True)):
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
tail -60 $O/r5b_pmc_summary.txt
