#!/bin/bash
# CRC32 kernel, chains per lane (csrc/crc.hip): the pipelined job of bench.py on a 48M-read shard with NGSQC_CRC_CHAINS = 1 (the round-2 kernel), 2, 4, each under
# rocprofv3 --kernel-trace --stats (the kernel's own duration next to the job's Mreads/s). Run on the GPU box from the repo root; writes gpurun_out/crc_ab.txt,
# which is committed as profiles/r05_crc_chains.txt.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
export NGSQC_BENCH_NO_STRONG=1 NGSQC_BENCH_NO_E2E=1 NGSQC_BENCH_NO_TOOLS=1 NGSQC_BENCH_NO_ONT=1
CMD="python $R/bench.py --reads 48000000 --steps 6 --warmup 2 --no-cpu-baseline --image-cache /dev/shm/ngsqc_ab_48m.bam"
cd /tmp && export TMPDIR=/tmp
$CMD > /dev/null 2>&1   # (generates the image, pages the python stack in)
: > $O/crc_ab.txt
for k in 1 4 2 4; do
	rm -rf $O/crc_ab_$k
	NGSQC_CRC_CHAINS=$k timeout 300 rocprofv3 --kernel-trace --stats -d $O/crc_ab_$k -o t -- $CMD > $O/crc_ab_$k.log 2>&1
	python - <<PY >> $O/crc_ab.txt
import sqlite3, glob, json
val = None
for ln in open("$O/crc_ab_$k.log"):
    if ln.startswith("{") and '"metric"' in ln:
        d = json.loads(ln); val = (d["value"], d["ms_per_step"])
print("NGSQC_CRC_CHAINS=$k\tjob %s Mreads/s, %s ms per step" % (val if val else ("?", "?")))
for db in glob.glob("$O/crc_ab_$k/**/*_results.db", recursive=True):
    for r in sqlite3.connect(db).execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        if any(x in r[0] for x in ("crc32", "huff_tokens", "lz77_groups")):
            print("\t%s\tcalls %d\ttotal %.3f ms\tavg %.4f ms\t%.2f %%" % (r[0].split("(")[0][-40:], r[1], r[2] / 1e3, r[3] / 1e3, r[4]))
PY
done
cat $O/crc_ab.txt
