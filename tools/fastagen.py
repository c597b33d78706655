#!/usr/bin/env python3
"""Synthetic reference genome (FASTA + .fai) for the GC / AT-dropout legs of the mapping QC (tests and bench; not product code).

The reference needs hg19 / hg38 for `AT dropout` / `GC dropout` (Statistics.cpp:363-387, 576-604); no genome ships with this repository, so
the tests use a synthetic one with the contig names and lengths of the BAM under test. Only the windows a test reads carry bases: the file
is written SPARSE (seek + write), everything else is a hole that reads back as NUL bytes, which Sequence::gcContent ignores like any other
non-ACGT character. The GC fraction of every 100-bp window is drawn from a seeded generator and covers the special cases: windows without
any A/C/G/T (bin -1), pure G/C windows (bin 100), pure A/T windows (bin 0).

usage: fastagen.py OUT.fa BAM_HEADER_TSV BED [seed]      (BAM_HEADER_TSV: name<TAB>length per line)
"""
import os
import sys

import numpy as np

LINE = 60   # bases per FASTA line


def write_sparse_fasta(path, refs, windows, seed=7, n_fill=("chrN", 0)):
    """refs: [(name, length)], windows: [(name, start1, end1)] 1-based closed intervals that must carry bases (padded by one FASTA line on both
    sides so that the reference's newline arithmetic, FastaFileIndex.cpp:96, stays inside written data). Returns {name: bytearray-free}."""
    rng = np.random.default_rng(seed)
    offsets = {}
    with open(path, "wb") as f, open(path + ".fai", "w") as fai:
        pos = 0
        for name, length in refs:
            hdr = f">{name}\n".encode()
            f.seek(pos); f.write(hdr); pos += len(hdr)
            offsets[name] = pos
            fai.write(f"{name}\t{length}\t{pos}\t{LINE}\t{LINE + 1}\n")
            n_lines = (length + LINE - 1) // LINE
            end = pos + length + n_lines
            f.seek(end - 1); f.write(b"\n")   # the contig's last byte: everything in between is a hole
            pos = end
        # bases of the requested windows: every 100-bp piece gets its own drawn GC fraction / special kind
        by_name = dict(refs)
        GC, AT = np.frombuffer(b"GC", dtype=np.uint8), np.frombuffer(b"AT", dtype=np.uint8)
        for name, s1, e1 in windows:
            lo = max(1, s1 - 2 * LINE); hi = min(by_name[name], e1 + 2 * LINE)
            n = hi - lo + 1
            if n <= 0:
                continue
            n_p = (n + 99) // 100
            u = np.repeat(rng.random(n_p), 100)[:n]          # GC fraction of the piece
            kind = np.repeat(rng.random(n_p), 100)[:n]       # < 0.03: no A/C/G/T, < 0.06: pure G/C, < 0.09: pure A/T
            soft = np.repeat(rng.random(n_p) < 0.2, 100)[:n]  # soft-masked (lower case) pieces count like upper case
            is_gc = np.where(kind < 0.06, True, np.where(kind < 0.09, False, rng.random(n) < u))
            seq = np.where(is_gc, rng.choice(GC, n), rng.choice(AT, n)).astype(np.uint8)
            seq = np.where(soft & (rng.random(n) < 0.5), seq | 0x20, seq)
            seq = np.where(kind < 0.03, np.uint8(ord("N")), seq).astype(np.uint8)
            # file image of bases [lo, hi]: a newline behind every LINE-th base of the contig
            b0 = np.arange(lo - 1, hi, dtype=np.int64)              # 0-based base indices
            off = b0 + b0 // LINE                                   # offsets relative to the contig's first base
            img = np.full(int(off[-1] - off[0]) + 1, ord("\n"), dtype=np.uint8)
            img[off - off[0]] = seq
            f.seek(offsets[name] + int(off[0])); f.write(img.tobytes())
    return path


def main():
    out, hdr, bed = sys.argv[1:4]
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 7
    refs = [(ln.split("\t")[0], int(ln.split("\t")[1])) for ln in open(hdr).read().splitlines() if ln.strip()]
    wins = []
    for ln in open(bed):
        if ln.startswith(("#", "track", "browser")) or not ln.strip():
            continue
        c, s, e = ln.split("\t")[:3]
        wins.append((c, int(s) + 1, int(e)))
    write_sparse_fasta(out, refs, wins, seed)
    print(out, os.path.getsize(out), "bytes (sparse)")


if __name__ == "__main__":
    main()
