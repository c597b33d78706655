#!/bin/bash
# First GPU call of the next round (≈8 GPU-minutes): the three prepared experiments of DESIGN.md section 7.
#   1. schedule: the fused job with NGSQC_K1_PHASED (a tile's decoder launches together, then its phase-2 launches alone) against the default, same image
#   2. end to end: the first job racing the background H2D copy (member table walked in pieces), incl. a slowed-down copy so that chunks really wait for pieces
#   3. instruction cache: SQC hit rate and fetch stalls of the pipelined job against the serialized one (five kernels share a CU pair's 64 KB when pipelined)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_probe; mkdir -p $O
python $R/tools/dev/job_probe.py 96000000 5 default,phased,phased_t4,default,phased,walk8 > $O/job_probe.log 2>&1; tail -7 $O/job_probe.log
python $R/tools/dev/race_probe.py 48000000 > $O/race_probe.log 2>&1; tail -4 $O/race_probe.log
export NGSQC_BENCH_NO_STRONG=1 NGSQC_BENCH_NO_E2E=1
CMD="python $R/bench.py --reads 48000000 --steps 3 --warmup 1 --no-cpu-baseline --image-cache /tmp/ngsqc_prof_48m.bam"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $O/icache_pipelined -o p -- $CMD > $O/icache_pipelined.log 2>&1
NGSQC_K1_SERIAL=1 NGSQC_PIPELINE=0 timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $O/icache_serial -o s -- $CMD > $O/icache_serial.log 2>&1
cd $R
python - <<'PY'
import glob, sqlite3
for mode in ("pipelined", "serial"):
    for p in glob.glob(f"gpurun_out/r4_probe/icache_{mode}/*.db"):
        con = sqlite3.connect(p); cur = con.cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
        pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")]; kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]; info = [t for t in tabs if "info_pmc" in t][0]
        q = f"select s.kernel_name, i.name, sum(e.value) from {pmc[0]} e join {info} i on e.pmc_id = i.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by 1, 2"
        acc = {}
        try:
            for k, c, v in cur.execute(q):
                nm = next((x for x in ("huff_tokens", "lz77_groups", "crc32", "walk_scan", "pileup", "scan_long") if x in k), k[:28])
                acc.setdefault(nm, {}); acc[nm][c] = acc[nm].get(c, 0) + v
        except Exception as e:
            print(mode, "query failed:", e, tabs[:6]); continue
        for k, d in sorted(acc.items()):
            if d.get("SQC_ICACHE_REQ", 0) > 1e6:
                print(f"{mode:10s} {k:28s} icache hit rate {d.get('SQC_ICACHE_HITS', 0) / d['SQC_ICACHE_REQ']:.4f}  ifetch level/fetch {d.get('SQ_IFETCH_LEVEL', 0) / max(d.get('SQ_IFETCH', 1), 1):.2f}  wait_inst share {d.get('SQ_WAIT_INST_ANY', 0) / max(d.get('SQ_WAVE_CYCLES', 1), 1):.3f}")
PY
