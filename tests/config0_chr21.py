"""BASELINE.json configs[0], literally: "MappingQC on 100k-read chr21 exome-subset BAM", as SURVEY.md 8(d) config 1 spells it out: 100 000 reads on chr21 (2x150 bp
paired end), ROI = about 2 000 exon-like intervals (length ~ LogNormal(mu 5.0, sigma 0.6) clipped to [60, 2000]) on 1 - 46.7 Mb. The reference checkout holds no such
file (its MappingQC inputs are panel-sized), so the instance is generated (tools/bamgen.cpp at 0.33x over the chromosome). Two users: tests/test_cpu_config0.py (the
CPU reference path = the oracle, held against the plain numpy restatement below; no GPU) and tests/test_gpu_config0.py (the MappingQC binary and the C ABI against
the oracle)."""
import numpy as np

import bamgen_lib as G

N_READS = 100_000
CHR21 = 20            # tid of chr21 in the generator's hg38 header
DEPTH = 0.33          # 100 000 x 150 bp over 45.5 of chr21's 46.7 Mb
N_EXONS = 2_000


def write_inputs(d):
    bam, bed = str(d / "chr21_100k.bam"), str(d / "chr21_exome_subset.bed")
    G.write(bam, n_reads=N_READS, seed=210, first_contig=CHR21, start_pos=0, depth=DEPTH)
    rng = np.random.default_rng(21)
    length = np.clip(np.exp(rng.normal(5.0, 0.6, N_EXONS)), 60, 2000).astype(np.int64)
    start = np.sort(rng.integers(1_000_000, 46_700_000 - 2_000, N_EXONS))          # (sorted for the eye only: some overlap, some touch; MappingQC merges its ROI)
    lines = [("chr21", int(s), int(s + n), "ex%04d" % i) for i, (s, n) in enumerate(zip(start, length))]
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

    def touches(lo, hi):   # does [lo, hi] overlap a line of the merged, sorted ROI: the first line that ends at or behind lo is the only candidate
        j = np.searchsorted(roi[:, 1], lo, side="left")
        return (j < len(roi)) & (roi[np.minimum(j, len(roi) - 1), 0] <= hi)
    near = on21 & touches(start - 250, end + 250)                          # :458-461
    on = on21 & touches(start, end)                                        # :464-467
    usable = on & ((flag & 0x400) == 0) & (mapq >= min_mapq)               # :478
    off = np.concatenate([[0], np.cumsum(roi[:, 1] - roi[:, 0] + 1)])      # where a line's bases begin in the depth array
    depth = np.zeros(int(off[-1]), dtype=np.int64)
    for s0, e0 in zip(start[usable], end[usable]):                         # (a few thousand reads touch the target)
        for j in range(int(np.searchsorted(roi[:, 1], s0, side="left")), len(roi)):
            if roi[j, 0] > e0:
                break
            lo, hi = max(s0, roi[j, 0]), min(e0, roi[j, 1])
            depth[off[j] + lo - roi[j, 0]: off[j] + hi - roi[j, 0] + 1] += 1
    proper = keep & ((flag & 1) != 0) & ((flag & 2) != 0)                # :520
    ins = proper & ~(mapped & spliced) & (np.abs(isize) < 1000)          # :524-534 (an unmapped read has no CIGAR that could splice)
    return {
        "al_total": int(keep.sum()), "al_mapped": int(mapped.sum()), "al_ontarget": int(on.sum()), "al_neartarget": int(near.sum()),
        "al_dup": int((keep & ((flag & 0x400) != 0)).sum()), "al_proper_paired": int(proper.sum()),
        "insert_size_read_count": int(ins.sum()), "insert_size_sum": int(np.abs(isize[ins]).sum()),
        "bases_mapped": int(l_seq[mapped].sum()), "bases_clipped": int(clipped[mapped].sum()),
        "bases_usable": int(depth.sum()), "max_length": int(l_seq[keep].max()), "paired_end": int(((flag[keep] & 1) != 0).any()),
    }, depth
