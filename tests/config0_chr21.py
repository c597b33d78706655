"""BASELINE.json configs[0], literally: "MappingQC on 100k-read chr21 exome-subset BAM". The reference checkout holds no such file (its MappingQC inputs are
panel-sized), so the instance is generated: 100 000 short reads (tools/bamgen.cpp, 2x150 bp, 30x) on chr21 from 14.0 Mb on, and an exome-like BED on the same
window. Two users: tests/test_cpu_config0.py (the CPU reference path = the oracle, held against the plain numpy restatement below; no GPU) and
tests/test_gpu_config0.py (the MappingQC binary and the C ABI against the oracle)."""
import numpy as np

import bamgen_lib as G

N_READS = 100_000
CHR21 = 20            # tid of chr21 in the generator's hg38 header
START = 14_000_000    # 100 000 x 150 bp at 30x = 500 kb of chr21 from here


def write_inputs(d):
    bam, bed = str(d / "chr21_100k.bam"), str(d / "chr21_exome_subset.bed")
    G.write(bam, n_reads=N_READS, seed=210, first_contig=CHR21, start_pos=START)
    rng = np.random.default_rng(21)
    lines, p = [], START + 2_000
    while p < START + 495_000:                      # exons of 60 .. 400 bp, introns of 0.4 .. 6 kb; every tenth exon is followed by one that overlaps it
        n = int(rng.integers(60, 400)); lines.append(("chr21", p, p + n, "ex%03d" % len(lines)))
        if len(lines) % 10 == 0:
            lines.append(("chr21", p + n // 2, p + n + 40, "ex%03d" % len(lines)))
        p += n + int(rng.integers(400, 6000))
    lines += [("chr1", 65_000, 65_600, "far1"), ("chrX", 2_800_000, 2_800_300, "far2")]   # targets without reads
    open(bed, "w").write("".join("%s\t%d\t%d\t%s\n" % ln for ln in lines))
    return bam, bed


def merged_chr21(bed):
    """the chr21 lines of the BED merged the way MappingQC merges its ROI (BedFile::merge: overlapping AND book-ended lines join), 1-based inclusive"""
    iv = sorted((int(f[1]) + 1, int(f[2])) for f in (ln.split("\t") for ln in open(bed)) if f[0] == "chr21")
    out = [list(iv[0])]
    for s, e in iv[1:]:
        if s <= out[-1][1] + 1:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return np.array(out, dtype=np.int64)


def restate(inflated, offsets, bed, min_mapq=1):
    """Statistics::mapping(bed_file, bam_file, ...) (src/cppNGS/Statistics.cpp:419-540) restated over arrays: one pass of numpy per counter instead of the reference's
    loop per read (and of the oracle's, oracle/stats.hpp mapping_roi). Only what configs[0] asks of the plumbing: the read counters, the usable bases and the per-base
    depth of the target region."""
    a = np.frombuffer(inflated, dtype=np.uint8)
    o = np.asarray(offsets, dtype=np.int64)

    def field(at, dt):
        w = np.dtype(dt).itemsize
        return np.ascontiguousarray(a[(o + at)[:, None] + np.arange(w)]).view(dt)[:, 0].astype(np.int64)
    tid, pos0, l_name, mapq = field(4, "<i4"), field(8, "<i4"), field(12, "u1"), field(13, "u1")
    n_cig, flag, l_seq, isize = field(16, "<u2"), field(18, "<u2"), field(20, "<i4"), field(32, "<i4")
    ref_len = np.zeros(len(o), dtype=np.int64); clipped = np.zeros(len(o), dtype=np.int64); spliced = np.zeros(len(o), dtype=bool)
    cig0 = o + 36 + l_name
    for k in range(int(n_cig.max())):               # the k-th operation of every read that has one
        m = n_cig > k
        w = np.ascontiguousarray(a[(cig0[m] + 4 * k)[:, None] + np.arange(4)]).view("<u4")[:, 0].astype(np.int64)
        op, ln = w & 15, w >> 4
        ref_len[m] += np.where(np.isin(op, (0, 2, 3, 7, 8)), ln, 0)     # M D N = X consume the reference (SAM spec 1.4.6)
        clipped[m] += np.where(np.isin(op, (4, 5)), ln, 0)              # Statistics.cpp:446-449
        spliced[m] |= op == 3
    keep = (flag & 0x900) == 0                                           # :419 secondary / supplementary alignments are not counted at all
    mapped = keep & ((flag & 4) == 0)
    start, end = pos0 + 1, pos0 + np.maximum(ref_len, 1)                 # 1-based, inclusive (BamAlignment::start / end)
    roi = merged_chr21(bed)
    on21 = mapped & (tid == CHR21)
    near = on21 & ((start[:, None] - 250 <= roi[None, :, 1]) & (end[:, None] + 250 >= roi[None, :, 0])).any(axis=1)     # :458-461
    ovl = (start[:, None] <= roi[None, :, 1]) & (end[:, None] >= roi[None, :, 0])
    on = on21 & ovl.any(axis=1)                                                                                            # :464-467
    usable = on & ((flag & 0x400) == 0) & (mapq >= min_mapq)                                                               # :478
    diff = np.zeros(int(roi[-1, 1]) - START + 2, dtype=np.int64)         # a difference array over the window, cut to the target afterwards
    np.add.at(diff, start[usable] - START, 1); np.add.at(diff, end[usable] + 1 - START, -1)
    cover = np.cumsum(diff)
    depth = np.concatenate([cover[s - START:e + 1 - START] for s, e in roi])
    proper = keep & ((flag & 1) != 0) & ((flag & 2) != 0)                # :520
    ins = proper & ~(mapped & spliced) & (np.abs(isize) < 1000)          # :524-534 (an unmapped read has no CIGAR that could splice)
    return {
        "al_total": int(keep.sum()), "al_mapped": int(mapped.sum()), "al_ontarget": int(on.sum()), "al_neartarget": int(near.sum()),
        "al_dup": int((keep & ((flag & 0x400) != 0)).sum()), "al_proper_paired": int(proper.sum()),
        "insert_size_read_count": int(ins.sum()), "insert_size_sum": int(np.abs(isize[ins]).sum()),
        "bases_mapped": int(l_seq[mapped].sum()), "bases_clipped": int(clipped[mapped].sum()),
        "bases_usable": int(depth.sum()), "max_length": int(l_seq[keep].max()), "paired_end": int(((flag[keep] & 1) != 0).any()),
    }, depth
