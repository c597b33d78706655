"""Turns the rocprofv3 sqlite outputs under gpurun_out/ into the text summaries committed under profiles/."""
import json
import sqlite3
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = open(f"profiles/{tag}_kernel_stats.txt", "w")
c = sqlite3.connect("gpurun_out/r1_trace/t_results.db")
out.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline   (48M-read shard per step, 1x MI355X)\n")
out.write("# name\tcalls\ttotal_ms\tavg_ms\tpercent\n")
for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    out.write(f"{r[0]}\t{r[1]}\t{r[2]/1e3:.3f}\t{r[3]/1e3:.4f}\t{r[4]:.2f}\n")
out.close()
pm = open(f"profiles/{tag}_hbm_traffic_pmc.txt", "w")
pm.write("# separate passes: rocprofv3 --pmc FETCH_SIZE -- <cmd> ; rocprofv3 --pmc WRITE_SIZE -- <cmd>  (same command as above)\n")
pm.write("# per-kernel SUM over dispatches / number of dispatches = per-launch value; rocprofv3 reports these in KiB.\n")
pm.write("# gfx950 note (MI355X_MICROARCH.md HBM section): FETCH_SIZE counts 64 B per 128 B request for wide coalesced streams (x2 correction);\n")
pm.write("# other access widths (the scan kernel's sparse 4-byte gathers, the inflate kernels' byte traffic) are uncalibrated.\n")
pm.write("# kernel\tcounter\tdispatches\tsum\tper_launch\n")
res = {}
for db, ctr in (("gpurun_out/r1_fetch/f_results.db", "FETCH_SIZE"), ("gpurun_out/r1_write/w_results.db", "WRITE_SIZE")):
    c = sqlite3.connect(db)
    for r in c.execute("select kernel_name, count(*), sum(value), max(value) from counters_collection where counter_name=? group by kernel_name", (ctr,)):
        pm.write(f"{r[0][:70]}\t{ctr}\t{r[1]}\t{r[2]:.1f}\tmax_launch={r[3]:.1f}\n")
        res[(r[0][:30], ctr)] = (r[1], r[2], r[3])
pm.close()
print(open(f"profiles/{tag}_kernel_stats.txt").read()[:1500])
print(open(f"profiles/{tag}_hbm_traffic_pmc.txt").read()[:3000])
