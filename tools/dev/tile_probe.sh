#!/bin/bash
# dev: value and un-pipelined scan-stage fraction of the MappingQC job for 1 / 2 / 4 K1 chunks per tile (192 M reads = 12 chunks)
mkdir -p gpurun_out/r3_tile; C=/dev/shm/ngsqc_tile_probe.bam
export NGSQC_BENCH_NO_STRONG=1 NGSQC_BENCH_NO_E2E=1
for t in 2 1 4 2; do
  NGSQC_TILE_CHUNKS=$t python bench.py --reads 192000000 --steps 4 --warmup 1 --no-cpu-baseline --image-cache $C > gpurun_out/r3_tile/t$t.json 2> gpurun_out/r3_tile/t$t.err
  python - $t <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r3_tile/t{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("chunks/tile", sys.argv[1], "tiles", d["config"]["tiles"], "value", d["value"], "scan frac", d["roofline_scan"]["frac"], "t_scan", d["roofline_scan"]["t_scan_ms"], "kernels", d["roofline_scan"]["scan_kernel_only"]["ms"], "K1 wall", d["stage_ms"]["inflate_stage_wall"])
PY
done
rm -f $C
