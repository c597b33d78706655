"""Hand-derived vectors for BedLowCoverage / BedHighCoverage and `-min_baseq`.

Every reference vector of these rows needs `panel.bam`, a blob that is missing from the reference checkout, so the oracle's restatement of
  BamAlignment::qualities              src/cppNGS/BamReader.cpp:210-255
  WorkerLowOrHighCoverage::run         src/cppNGS/WorkerLowOrHighCoverage.cpp:18-107      (random access: int depth per line)
  WorkerLowOrHighCoverageChr::run      src/cppNGS/WorkerLowOrHighCoverage.cpp:141-235     (sweep: unsigned char depth, `if (cov[p]<254) ++cov[p]`)
  Statistics::lowOrHighCoverage        src/cppNGS/Statistics.cpp:2534-2657                (output.merge(true, true, true))
  BedLowCoverage main                  src/BedLowCoverage/main.cpp:55-57                  (file.merge(true, true) of the input first)
had no witness that the builder's own code did not produce. The cases below are tiny BAMs whose expected per-base depth and output BED lines are WRITTEN OUT BY
HAND from those reference lines (the derivation is in each case's comment); tests/test_oracle_golden.py checks the oracle against them, tests/test_gpu_lowhigh.py
the product (C ABI and tool bytes). Nothing here is computed."""
import struct
import zlib

REFS = [("chr1", 10000)]
TEXT = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:10000\n"
OPS = "MIDNSHP=X"


def record(name, pos1, cigar, quals=None, flag=0, mapq=60, tid=0):
    """one BAM record (SAM spec 4.2): pos1 = 1-based start, cigar = [(op, len)], quals = list of phred values (default 40 for every base)"""
    l_seq = sum(k for op, k in cigar if op in "MIS=X")
    quals = bytes([40] * l_seq if quals is None else quals)
    assert len(quals) == l_seq
    nm = name.encode() + b"\0"
    cig = b"".join(struct.pack("<I", k << 4 | OPS.index(op)) for op, k in cigar)
    seq = bytes([0x11] * ((l_seq + 1) // 2))   # AAAA...
    body = struct.pack("<iiBBHHHiiii", tid, pos1 - 1, len(nm), mapq, 4680, len(cigar), flag, l_seq, -1, -1, 0) + nm + cig + seq + quals
    return struct.pack("<i", len(body)) + body


def _bgzf(raw):
    c = zlib.compressobj(6, zlib.DEFLATED, -15); body = c.compress(raw) + c.flush()
    return b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body + struct.pack("<II", zlib.crc32(raw), len(raw))


def write_bam(path, records):
    hdr = b"BAM\1" + struct.pack("<i", len(TEXT)) + TEXT.encode() + struct.pack("<i", len(REFS))
    for name, ln in REFS:
        hdr += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
    out = _bgzf(hdr)
    chunk = b""
    for r in records:
        if len(chunk) + len(r) > 60000:
            out += _bgzf(chunk); chunk = b""
        chunk += r
    if chunk:
        out += _bgzf(chunk)
    open(path, "wb").write(out + _bgzf(b""))


# ---- the reads (sorted by position) ----
READS = []
# C1 / C2: three reads of 10M at 101, 106, 111 - and five reads that every coverage worker skips (WorkerLowOrHighCoverage.cpp:47-49): duplicate, secondary,
# supplementary, unmapped, MAPQ 0 < min_mapq 1
READS += [record("c1_a", 101, [("M", 10)]), record("c2_dup", 101, [("M", 20)], flag=0x400), record("c2_sec", 101, [("M", 20)], flag=0x100),
          record("c2_sup", 101, [("M", 20)], flag=0x800), record("c2_unm", 101, [("M", 20)], flag=0x4), record("c2_mq0", 101, [("M", 20)], mapq=0),
          record("c1_b", 106, [("M", 10)]), record("c1_c", 111, [("M", 10)])]
# C3: 5= 5M at 201, qualities low at read index 0 and 5. BamAlignment::qualities has no branch for '=' / 'X' (BamReader.cpp:225-253): neither index moves, so the
# M operation reads q[0..4] (not q[5..9]) and clears genome bits 0..4 (not 5..9): q[0] = 5 < 20 clears bit 0 = position 201; q[5] is never looked at.
READS += [record("c3", 201, [("=", 5), ("M", 5)], quals=[5, 40, 40, 40, 40, 5, 40, 40, 40, 40])]
# C4: 2S 3M 2I 3M 2D 2M 3N 2M at 301: read indices S 0-1, M 2-4, I 5-6, M 7-9, M 10-11, M 12-13; genome offsets M 0-2, M 3-5, D 6-7, M 8-9, N 10-12, M 13-14
# (reference span 301..315). Low qualities at read index 3 (genome 1 = 302), 5 (an inserted base: never tested), 8 (genome 4 = 305), 12 (genome 13 = 314).
# S and I advance the read index only (:240-247), D and N the genome index only (:236-239,:244-247): their positions STAY true = covered.
READS += [record("c4", 301, [("S", 2), ("M", 3), ("I", 2), ("M", 3), ("D", 2), ("M", 2), ("N", 3), ("M", 2)],
                 quals=[40, 40, 40, 5, 40, 5, 40, 40, 5, 40, 40, 40, 5, 40])]
# C5: 300 reads of 10M at 401: int depth 300 per line in random access (WorkerLowOrHighCoverage.cpp:39,70-73), unsigned char that stops at 254 in the sweep (:184-196)
READS += [record("c5_%03d" % i, 401, [("M", 10)]) for i in range(300)]
# C7: two reads of 10M at 611 inside a line 601..630
READS += [record("c7_a", 611, [("M", 10)]), record("c7_b", 611, [("M", 10)])]

# ---- the cases: BED lines (0-based start, end, name), parameters, expected per-base depth of the (merged) lines, expected output lines ----
# expected["tool"]: what BedLowCoverage / BedHighCoverage print (input merged with merge(true, true) first); expected["function"]: Statistics::lowOrHighCoverage
# on the lines as they are (the unit-test level); each for random access (ra) and the sweep (sw) when they differ
CASES = {
    # depth: 101-105 one read, 106-110 two (a + b), 111-115 two (b + c), 116-120 one. cutoff 2: low = depth < 2 (:84-86), high = depth >= 2 (:80)
    "c1_low": dict(bed=[(100, 120, "L1")], cutoff=2, is_high=False, min_baseq=0, depth=[1] * 5 + [2] * 10 + [1] * 5,
                   out=["chr1\t100\t105\tL1", "chr1\t115\t120\tL1"]),
    "c1_high": dict(bed=[(100, 120, "L1")], cutoff=2, is_high=True, min_baseq=0, depth=[1] * 5 + [2] * 10 + [1] * 5, out=["chr1\t105\t115\tL1"]),
    # min_baseq 20: position 201 is masked (the quirk above), 202..210 counted; without min_baseq all ten
    "c3_baseq": dict(bed=[(200, 210, "L2")], cutoff=1, is_high=False, min_baseq=20, depth=[0] + [1] * 9, out=["chr1\t200\t201\tL2"]),
    "c3_plain": dict(bed=[(200, 210, "L2")], cutoff=1, is_high=False, min_baseq=0, depth=[1] * 10, out=[]),
    # 301..315: 302, 305, 314 masked; deleted (307, 308) and skipped (311..313) positions covered
    "c4_baseq": dict(bed=[(300, 315, "L3")], cutoff=1, is_high=False, min_baseq=20, depth=[1, 0, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1],
                     out=["chr1\t301\t302\tL3", "chr1\t304\t305\tL3", "chr1\t313\t314\tL3"]),
    "c4_plain": dict(bed=[(300, 315, "L3")], cutoff=1, is_high=False, min_baseq=0, depth=[1] * 15, out=[]),
    # 300 reads: high at cutoff 255 only in random access (sweep: 254 < 255 everywhere); low at cutoff 255 only in the sweep
    "c5_high": dict(bed=[(400, 410, "L4")], cutoff=255, is_high=True, min_baseq=0, depth=[300] * 10, out_ra=["chr1\t400\t410\tL4"], out_sw=[]),
    "c5_low": dict(bed=[(400, 410, "L4")], cutoff=255, is_high=False, min_baseq=0, depth=[300] * 10, out_ra=[], out_sw=["chr1\t400\t410\tL4"]),
    # three back-to-back lines without reads. The TOOL merges its input with merge(true, true) (names joined, not made unique: BedFile.cpp:287-295 with
    # merged_names_unique = false): one line "gA,gB,gA", printed as it is. The FUNCTION gets three lines, emits three runs and joins them with
    # merge(true, true, true) (Statistics.cpp:2655): adjacent runs merge, a name that is already there is not added again: "gA,gB"
    "c6_names": dict(bed=[(500, 510, "gA"), (510, 520, "gB"), (520, 530, "gA")], cutoff=1, is_high=False, min_baseq=0, depth=[0] * 30,
                     out=["chr1\t500\t530\tgA,gB,gA"], out_function=["chr1\t500\t530\tgA,gB"]),
    # two reads at 611..620 in the line 601..630, cutoff 2: the runs 601..610 and 621..630 do not touch, both keep the line's name
    "c7_two_runs": dict(bed=[(600, 630, "gC")], cutoff=2, is_high=False, min_baseq=0, depth=[0] * 10 + [2] * 10 + [0] * 10, out=["chr1\t600\t610\tgC", "chr1\t620\t630\tgC"]),
}


def expected(case, random_access, function_level=False):
    c = CASES[case]
    if function_level and "out_function" in c:
        return c["out_function"]
    if "out" in c:
        return c["out"]
    return c["out_ra"] if random_access else c["out_sw"]


def write_bed(path, case):
    open(path, "w").write("".join("chr1\t%d\t%d\t%s\n" % ln for ln in CASES[case]["bed"]))
