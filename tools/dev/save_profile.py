"""Turns the rocprofv3 sqlite outputs under gpurun_out/ into the text summaries committed under profiles/.

  rocprofv3 --kernel-trace --stats -d gpurun_out/r1_trace -o t -- <cmd>
  rocprofv3 --pmc FETCH_SIZE        -d gpurun_out/r1_fetch -o f -- <cmd>      (separate passes, no trace domains)
  rocprofv3 --pmc WRITE_SIZE        -d gpurun_out/r1_write -o w -- <cmd>
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY -d gpurun_out/r1_sq -o s -- <cmd>
  <cmd> = python bench.py --steps 3 --warmup 1 --no-cpu-baseline
"""
import os
import sqlite3
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
PFX = "r1" if tag.startswith("r01") else ("r2" if tag.startswith("r02") else ("r3" if tag.startswith("r03") else ("r4" if tag.startswith("r04") else ("r5" if tag.startswith("r05") else "r6"))))
# what the per-member / per-record figures depend on: bench.py only scales them to its own launches when it runs with the same settings
SETTINGS = "settings: k1_format=r06-word-per-trip-64B-lines tile_chunks=%s token_slots=%s" % (os.environ.get("NGSQC_TILE_CHUNKS", "8"), os.environ.get("NGSQC_TOKEN_SLOTS", "8"))
import math
CHUNKS_PER_JOB = math.ceil(249024 / (int(os.environ.get("NGSQC_K1_CHUNK_WAVES", "5")) * 256 * 64)) if PFX == "r6" else 3   # K1 launches per job of the 48 M-read shard (249 024 members)
try:
    import subprocess
    SETTINGS += " commit=" + subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    pass
CMD = ("python bench.py --steps 3 --warmup 1 --no-cpu-baseline   (48M-read shard per step, 1x MI355X)" if PFX == "r1" else
       "NGSQC_BENCH_NO_STRONG=1 NGSQC_BENCH_NO_E2E=1 python bench.py --reads 48000000 --steps 3 --warmup 1 --no-cpu-baseline   (48M-read shard; K1 chunks per job: 3 until round 5, 4 since round 6 (chunks of 5 waves per CU); 6 jobs per run: 1 warm-up + 3 timed + the un-pipelined and the isolated-K1 extra steps; 1x MI355X; counter passes with NGSQC_K1_SERIAL=1 NGSQC_PIPELINE=0: every kernel alone on the chip)" if PFX in ("r3", "r4", "r5", "r6") else
       "python bench.py --reads 48000000 --steps 3 --warmup 1 --no-cpu-baseline   (48M-read shard = 2 tiles per step, 1x MI355X; counter passes with NGSQC_K1_SERIAL=1 NGSQC_PIPELINE=0: every kernel alone on the chip)")
out = open(f"profiles/{tag}_kernel_stats.txt", "w")
c = sqlite3.connect(f"gpurun_out/{PFX}_trace/t_results.db")
out.write(f"# rocprofv3 --kernel-trace --stats -- {CMD}\n")
out.write("# name\tcalls\ttotal_ms\tavg_ms\tpercent\n")
for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    out.write(f"{r[0]}\t{r[1]}\t{r[2]/1e3:.3f}\t{r[3]/1e3:.4f}\t{r[4]:.2f}\n")
out.close()
pm = open(f"profiles/{tag}_hbm_traffic_pmc.txt", "w")
pm.write(f"# separate passes: rocprofv3 --pmc FETCH_SIZE -- <cmd> ; rocprofv3 --pmc WRITE_SIZE -- <cmd> ; <cmd> = {CMD}\n")
pm.write("# per-kernel SUM over dispatches / number of dispatches = per-launch value; rocprofv3 reports these in KiB.\n")
pm.write("# gfx950 note (MI355X_MICROARCH.md HBM section): FETCH_SIZE counts 64 B per 128 B request for wide coalesced streams (x2 correction);\n")
pm.write("# other access widths (the scan kernel's sparse 4-byte gathers, the inflate kernels' byte traffic) are uncalibrated.\n")
if PFX in ("r4", "r5", "r6"):
    pm.write(f"# {SETTINGS}\n")
pm.write("# kernel\tcounter\tdispatches\tsum_KiB\tper_launch\n")
for db, ctr in ((f"gpurun_out/{PFX}_fetch/f_results.db", "FETCH_SIZE"), (f"gpurun_out/{PFX}_write/w_results.db", "WRITE_SIZE")):
    if not os.path.exists(db):
        continue
    c = sqlite3.connect(db)
    for r in c.execute("select kernel_name, count(*), sum(value), max(value) from counters_collection where counter_name=? group by kernel_name", (ctr,)):
        pm.write(f"{r[0][:70]}\t{ctr}\t{r[1]}\t{r[2]:.1f}\tavg_launch={r[2]/r[1]:.1f}\n")
if PFX in ("r3", "r4", "r5", "r6"):
    # bytes per BGZF member (K1 kernels: 3 chunk launches of 83 008 members per job + one 8-member launch of the header read) and per record (K2 / scan
    # kernels: 48 000 000 records per job), raw counter x 1024 - what bench.py scales to its own launches (roofline.traffic, roofline_scan.traffic)
    pm.write("# kernel\tcounter\tbytes_per_member | bytes_per_record\tvalue\t(jobs in the run)\n")
    for db, ctr in ((f"gpurun_out/{PFX}_fetch/f_results.db", "FETCH_SIZE"), (f"gpurun_out/{PFX}_write/w_results.db", "WRITE_SIZE")):
        if not os.path.exists(db):
            continue
        c = sqlite3.connect(db)
        rows = {r[0]: (r[1], r[2]) for r in c.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name=? group by kernel_name", (ctr,))}
        jobs = None
        for k, (n, tot) in rows.items():
            if "huff_tokens_kernel" in k:
                jobs = (n - 1) / float(CHUNKS_PER_JOB)
        for k, (n, tot) in rows.items():
            short = k.split("(")[0].split("::")[-1].split("<")[0]
            if jobs and short in ("huff_tokens_kernel", "lz77_groups_kernel", "crc32_kernel", "crc32_chains_kernel"):
                pm.write(f"{short}\t{ctr}\tbytes_per_member\t{tot * 1024.0 / (jobs * 249024 + 8):.2f}\t{jobs:.1f}\n")
            elif jobs and short in ("walk_scan_kernel", "scan_kernel", "index_count_kernel", "index_write_kernel", "index_guess_kernel", "pileup_kernel"):
                pm.write(f"{short}\t{ctr}\tbytes_per_record\t{tot * 1024.0 / (jobs * 48000000):.3f}\t{jobs:.1f}\n")
pm.close()
if os.path.exists(f"gpurun_out/{PFX}_sq/s_results.db"):
    sq = open(f"profiles/{tag}_sq_counters.txt", "w")
    sq.write(f"# rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY -- {CMD}\n")
    sq.write("# SQ_INSTS_* count wave-level instructions, the *_CYCLES counters tick in quad-cycles (one wave64 VALU instruction = one tick);\n")
    sq.write("# derived: valu_issue_share = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (share of a wave's lifetime spent issuing VALU)\n")
    sq.write("# kernel\tcounter\tdispatches\tsum_over_dispatches\n")
    c = sqlite3.connect(f"gpurun_out/{PFX}_sq/s_results.db")
    acc = {}
    for k, cn, n, s in c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
        sq.write(f"{k[:60]}\t{cn}\t{n}\t{s:.4e}\n")
        acc.setdefault(k[:60], {})[cn] = s
    sq.write("# kernel\tvalu_issue_share\tany_issue_share\tvalu_per_salu\n")
    for k, v in acc.items():
        if v.get("SQ_WAVE_CYCLES"):
            sq.write(f"{k}\t{v.get('SQ_ACTIVE_INST_VALU', 0)/v['SQ_WAVE_CYCLES']:.3f}\t{v.get('SQ_ACTIVE_INST_ANY', 0)/v['SQ_WAVE_CYCLES']:.3f}\t{v.get('SQ_INSTS_VALU', 0)/max(v.get('SQ_INSTS_SALU', 1), 1):.2f}\n")
    sq.close()
# r02 extras: the kernel trace of the serialized run (every kernel alone) and the second SQ pass (stall / LDS counters)
if os.path.exists(f"gpurun_out/{PFX}_trace_serial/t_results.db"):
    o2 = open(f"profiles/{tag}_kernel_stats_serial.txt", "w")
    o2.write(f"# NGSQC_K1_SERIAL=1 NGSQC_PIPELINE=0 rocprofv3 --kernel-trace --stats -- {CMD}\n# name\tcalls\ttotal_ms\tavg_ms\tpercent\n")
    for r in sqlite3.connect(f"gpurun_out/{PFX}_trace_serial/t_results.db").execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        o2.write(f"{r[0]}\t{r[1]}\t{r[2]/1e3:.3f}\t{r[3]/1e3:.4f}\t{r[4]:.2f}\n")
    o2.close()
if os.path.exists(f"gpurun_out/{PFX}_sq2/s_results.db"):
    sq = open(f"profiles/{tag}_sq_stall_counters.txt", "w")
    sq.write(f"# rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -- {CMD}\n")
    sq.write("# WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles); LDS_BANK_CONFLICT = extra LDS cycles, LDS_IDX_ACTIVE = all LDS-array cycles\n")
    sq.write("# kernel\tcounter\tdispatches\tsum_over_dispatches\n")
    acc = {}
    for k, cn, n, sm in sqlite3.connect(f"gpurun_out/{PFX}_sq2/s_results.db").execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
        sq.write(f"{k[:60]}\t{cn}\t{n}\t{sm:.4e}\n"); acc.setdefault(k[:60], {})[cn] = sm
    sq.write("# kernel\twait_any_share\twait_inst_any_share\tlds_conflict_share_of_lds_cycles\n")
    for k, v in acc.items():
        if v.get("SQ_WAVE_CYCLES"):
            sq.write(f"{k}\t{v.get('SQ_WAIT_ANY', 0)/v['SQ_WAVE_CYCLES']:.3f}\t{v.get('SQ_WAIT_INST_ANY', 0)/v['SQ_WAVE_CYCLES']:.3f}\t{v.get('SQ_LDS_BANK_CONFLICT', 0)/max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f}\n")
    sq.close()
print(open(f"profiles/{tag}_kernel_stats.txt").read()[:1800])
print(open(f"profiles/{tag}_hbm_traffic_pmc.txt").read()[:3000])
