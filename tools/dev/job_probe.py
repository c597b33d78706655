#!/usr/bin/env python3
"""Dev probe: the fused MappingQC -wgs job (mapping_wgs + OMIM ROI + site pileup) on a synthetic shard under a few switches.
usage: job_probe.py [reads=48000000] [steps=5] [variants=default,nopipe,nocrc,...]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
ngsqc = importlib.import_module("ngs-bits_amd")
import bamgen_lib as G  # noqa: E402
import hostprep as H  # noqa: E402

VARIANTS = {
    "default": {},
    "phased": {"NGSQC_K1_PHASED": "1"}, "phased_t4": {"NGSQC_K1_PHASED": "1", "NGSQC_TILE_CHUNKS": "4", "NGSQC_TOKEN_SLOTS": "4"}, "walk8": {"NGSQC_WALK_THREADS": "8"},
    "nopipe": {"NGSQC_PIPELINE": "0"},
    "nocrc": {"NGSQC_VERIFY_CRC": "0"},
    "div2": {"NGSQC_K1_CHUNK_DIV": "2"},
    "div4": {"NGSQC_K1_CHUNK_DIV": "4"},
    "tile1": {"NGSQC_TILE_CHUNKS": "1"},
    "tile4": {"NGSQC_TILE_CHUNKS": "4"},
    "unsorted": {"NGSQC_K1_SORTED": "0"},
    "park8": {"NGSQC_P1_PARK": "8"}, "park24": {"NGSQC_P1_PARK": "24"}, "park32": {"NGSQC_P1_PARK": "32"},
    "debug": {"NGSQC_DEBUG": "1"},
    "p2wg768": {"NGSQC_P2_WGS": "768"}, "p2wg2048": {"NGSQC_P2_WGS": "2048"}, "p2wg512": {"NGSQC_P2_WGS": "512"},
    "p2wg3072": {"NGSQC_P2_WGS": "3072"}, "p2wg8192": {"NGSQC_P2_WGS": "8192"}, "p2wg16384": {"NGSQC_P2_WGS": "16384"}, "p2wg1536": {"NGSQC_P2_WGS": "1536"},
    "p2wg4096_nopad": {"NGSQC_P2_WGS": "4096", "NGSQC_P1_PAD": "0"},
    "p2wg1024": {"NGSQC_P2_WGS": "1024"}, "p2wg4096": {"NGSQC_P2_WGS": "4096"},
    "nopad": {"NGSQC_P1_PAD": "0"}, "nopad_1s": {"NGSQC_P1_PAD": "0", "NGSQC_P1_STREAMS": "1"}, "p1_1s": {"NGSQC_P1_STREAMS": "1"},
    "nopad_p2wg2048": {"NGSQC_P1_PAD": "0", "NGSQC_P2_WGS": "2048"}, "nopad_1s_p2wg2048": {"NGSQC_P1_PAD": "0", "NGSQC_P1_STREAMS": "1", "NGSQC_P2_WGS": "2048"},
    "prio0": {"NGSQC_P1_PRIO": "0"}, "prio1": {"NGSQC_P1_PRIO": "1"}, "prio3": {"NGSQC_P1_PRIO": "3"},
    "prio3_nopad": {"NGSQC_P1_PRIO": "3", "NGSQC_P1_PAD": "0"}, "prio0_nocrc": {"NGSQC_P1_PRIO": "0", "NGSQC_VERIFY_CRC": "0"}, "prio3_nocrc": {"NGSQC_P1_PRIO": "3", "NGSQC_VERIFY_CRC": "0"},
    "mul2": {"NGSQC_K1_CHUNK_MUL": "2", "NGSQC_TILE_CHUNKS": "1"}, "mul3": {"NGSQC_K1_CHUNK_MUL": "3", "NGSQC_TILE_CHUNKS": "1"}, "mul2_t2": {"NGSQC_K1_CHUNK_MUL": "2"},
    "serial": {"NGSQC_K1_SERIAL": "1", "NGSQC_PIPELINE": "0"},
    "dmasafe": {"NGSQC_P1_DMA_SAFE": "1"}, "serial_dmasafe": {"NGSQC_K1_SERIAL": "1", "NGSQC_PIPELINE": "0", "NGSQC_P1_DMA_SAFE": "1"},
    "bufs2": {"NGSQC_TILE_BUFFERS": "2", "NGSQC_TOKEN_SLOTS": "3"}, "slots3": {"NGSQC_TOKEN_SLOTS": "3"}, "bufs4": {"NGSQC_TILE_BUFFERS": "4", "NGSQC_TOKEN_SLOTS": "5"},
    "tile1_b4": {"NGSQC_TILE_CHUNKS": "1", "NGSQC_TILE_BUFFERS": "4", "NGSQC_TOKEN_SLOTS": "4"}, "tile1_b3": {"NGSQC_TILE_CHUNKS": "1"}, "tile3": {"NGSQC_TILE_CHUNKS": "3", "NGSQC_TOKEN_SLOTS": "5"},
    "prio3_park32": {"NGSQC_P1_PRIO": "3", "NGSQC_P1_PARK": "32"}, "prio2": {"NGSQC_P1_PRIO": "2"},
    "p2pad2k": {"NGSQC_P2_LDS_PAD": "2048"}, "p2pad6k": {"NGSQC_P2_LDS_PAD": "6144"}, "p2pad1k": {"NGSQC_P2_LDS_PAD": "1024"},
    "crcserial": {"NGSQC_CRC_STREAM": "0"}, "crcstream": {"NGSQC_CRC_STREAM": "1"},
    "park16": {"NGSQC_P1_PARK": "16"}, "park48": {"NGSQC_P1_PARK": "48"},
}


def main():
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 48_000_000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    names = sys.argv[3].split(",") if len(sys.argv) > 3 else ["default", "nopipe", "nocrc"]
    flavor = int(sys.argv[4]) if len(sys.argv) > 4 else 0; level = int(sys.argv[5]) if len(sys.argv) > 5 else 6
    t0 = time.time()
    cache = f'/tmp/ngsqc_probe_{reads}_{flavor}_{level}.bam'
    image = np.fromfile(cache, dtype=np.uint8) if os.path.exists(cache) else G.generate(reads, flavor=flavor, level=level)
    if not os.path.exists(cache):
        image.tofile(cache)
    print(f"[probe] flavor {flavor} level {level}: generated {reads} reads, {image.size} bytes in {time.time() - t0:.1f} s on {os.cpu_count()} cpus", flush=True)
    omim = os.path.join(ROOT, "ngs-bits_amd", "resources", "hg38_440_omim_genes.bed")
    ref = None
    for name in names:
        for k in list(os.environ):
            if k.startswith("NGSQC_"):
                del os.environ[k]
        if name in VARIANTS:
            os.environ.update(VARIANTS[name])
        else:   # ad-hoc: "serial+P1_WAVES=8+K1_CHUNK_MUL=2" (names of VARIANTS and NGSQC_ switches joined by +)
            for part in name.split("+"):
                if part in VARIANTS:
                    os.environ.update(VARIANTS[part])
                else:
                    k, v = part.split("="); os.environ["NGSQC_" + k] = v
        t0 = time.time()
        h = ngsqc.Handle(data=image, device=0)
        t_open = time.time() - t0
        refs = h.refs
        regs, _ = H.bed_regions(omim, refs, 3)
        tx, ty = H.xy_tids(refs)
        sites = H.known_sites(refs) if hasattr(H, "known_sites") else None
        mp = dict(mode=ngsqc.MODE_WGS, regions=regs, min_mapq=1, tid_x=tx, tid_y=ty, nonspecial=H.nonspecial(refs))
        walls = []
        for i in range(steps + 1):
            h.drop_decoded()
            t1 = time.perf_counter()
            out = h.run_job(mapping=mp, sites=sites)
            walls.append(1e3 * (time.perf_counter() - t1))
        tm = h.timings()
        c = out["counters"]
        if ref is None:
            ref = c.copy()
        same = bool(np.array_equal(ref, c))
        w = float(np.mean(walls[1:]))
        print(f"[probe] {name:28s} open {t_open:.2f}s wall {w:8.2f} ms  ({tm['n_records'] / w / 1e3:7.1f} Mreads/s)  K1 {tm['inflate_ms']:.2f} (huff {tm['inflate_huff_ms']:.1f} lz {tm['inflate_lz77_ms']:.1f} x{tm['inflate_huff_launches']}) "
              f"index {tm['index_ms']:.2f} scan {tm['scan_ms']:.2f} (kernels {tm['scan_kernel_ms']:.2f}) pile {tm['pileup_ms']:.2f} fin {tm['finalize_ms']:.2f} tiles {tm['n_tiles']} members {tm['members_inflated']}/{h.n_blocks} "
              f"records {tm['n_records']} same_counters {same}", flush=True)
        h.close()


if __name__ == "__main__":
    main()
