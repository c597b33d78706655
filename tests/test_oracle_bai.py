"""Pins oracle/bai_build.py (the CPU restatement of htslib's BAI construction, the checker of ngsqc_write_bai) on the reference's own fixtures: every
BAM under tests/golden/ref_in comes with the .bai that samtools / htslib wrote for it. No GPU."""
import glob
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GI = os.path.join(HERE, "golden", "ref_in")
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import bai_build as B  # noqa: E402

PAIRS = sorted(p[:-4] for p in glob.glob(os.path.join(GI, "*.bam.bai")))


def matching_variants(bam):
    exp, n_no_coor = B.parse_bai(bam + ".bai")
    return [v for v in B.VARIANTS if B.build_for_bam(bam, v) == (exp, n_no_coor)], n_no_coor


def test_fixture_indices_are_reproduced():
    assert len(PAIRS) >= 15
    current = 0; older = 0; pre_htslib = []
    for bam in PAIRS:
        vs, n_no_coor = matching_variants(bam)
        if n_no_coor is None:   # written by samtools 0.1.x (no n_no_coor field): not an htslib index
            pre_htslib.append(os.path.basename(bam)); continue
        assert vs, f"{os.path.basename(bam)}: the fixture index is not reproduced by any htslib variant"
        if B.CURRENT in vs: current += 1
        else: older += 1
    assert current >= 10, (current, older)
    assert sorted(pre_htslib) == ["lowcov_bug_case1.bam", "lowcov_bug_case2.bam", "sry.bam"]


def test_variants_differ_only_where_documented():
    """the linear index of a fixture with windows without reads separates the fill directions"""
    bam = os.path.join(GI, "BamReader_rna.bam")
    vs, _ = matching_variants(bam)
    assert vs == [B.CURRENT]
    bam = os.path.join(GI, "MappingQC_in3.bam")
    vs, _ = matching_variants(bam)
    assert vs == [("forward", "file_end")]


def test_reg2bin_known_answers():
    # SAM spec 5.3: the bins of [beg, end) at the five levels
    assert B.reg2bin(0, 1) == 4681 and B.reg2bin(16383, 16384) == 4681 and B.reg2bin(16383, 16385) == 585
    assert B.reg2bin(0, 1 << 29) == 0 and B.reg2bin((1 << 29) - 1, 1 << 29) == 37448
    assert B.reg2bin(-1, 0) == 4680   # a read without reference, as htslib computes it


@pytest.mark.parametrize("what", ["unsorted", "not_continuous", "no_coor_in_front"])
def test_rejects_what_htslib_rejects(what):
    recs = {"unsorted": [(0, 100, 150, 10, True), (0, 50, 100, 20, True)],
            "not_continuous": [(0, 100, 150, 10, True), (1, 5, 9, 20, True), (0, 200, 210, 30, True)],
            "no_coor_in_front": [(-1, -1, 0, 10, False), (0, 5, 9, 20, True)]}[what]
    with pytest.raises(ValueError):
        B.build(2, 0, recs, 40)


# ---- CSI (oracle/csi_build.py): pinned through the BAI fixtures - at BAI's geometry a CSI holds the same bins and chunks, and loff is the linear index at the bin's first window ----
import csi_build as CSI  # noqa: E402


def test_csi_at_bai_geometry_equals_the_fixture_indices():
    n = 0
    for bam in PAIRS:
        vs, n_no_coor = matching_variants(bam)
        if B.CURRENT not in vs: continue
        exp, _ = B.parse_bai(bam + ".bai")
        geom, refs, nnc = CSI.build_for_bam(bam, 14, 5)
        assert geom == (14, 5) and nnc == n_no_coor and len(refs) == len(exp)
        for bins, (ebins, lidx) in zip(refs, exp):
            assert {b: c for b, (_, c) in bins.items()} == ebins
            for b, (loff, _) in bins.items():
                if b == B.META_BIN: assert loff == 0; continue
                bot = B.bin_bot(b, geom)
                assert loff == (lidx[bot] if bot < len(lidx) else 0), (os.path.basename(bam), b)
        n += 1
    assert n >= 10


def test_csi_serialization_round_trip(tmp_path):
    bam = os.path.join(GI, "MappingQC_in2.bam")
    for min_shift, compress in [(14, True), (12, False), (16, True)]:
        csi = CSI.build_for_bam(bam, min_shift)
        p = str(tmp_path / f"x{min_shift}.csi"); CSI.write_csi(p, csi, compress)
        assert CSI.parse_csi(p) == csi
        assert open(p, "rb").read(2) == (b"\x1f\x8b" if compress else b"CS")


def test_csi_depth_rule():
    assert CSI.depth_for([248956422], 14) == 5        # human chr1: the BAI geometry
    assert CSI.depth_for([16569], 14) == 1            # chrM alone: 16569 + 256 > 2^14
    assert CSI.depth_for([16000], 14) == 0
    assert CSI.depth_for([(1 << 29) - 256], 14) == 5 and CSI.depth_for([(1 << 29) - 255], 14) == 6
    assert CSI.depth_for([], 14) == 0
    assert CSI.depth_for([248956422], 12) == 6


@pytest.mark.parametrize("name,min_shift", [("MappingQC_in2.bam", 14), ("BamReader_rna.bam", 12), ("close_exons.bam", 17), ("Statistics_longread.bam", 10)])
def test_csi_query_holds_every_overlapping_record(name, min_shift):
    import random
    bam = os.path.join(GI, name)
    n_ref, offset0, recs, final = B.read_bam(bam)
    csi = CSI.build_for_bam(bam, min_shift)
    starts = [offset0] + [r[3] for r in recs[:-1]]
    mapped = [i for i, r in enumerate(recs) if r[0] >= 0]
    rng = random.Random(5)
    for _ in range(80):
        i = rng.choice(mapped); tid, p0, e0 = recs[i][:3]
        beg = max(0, p0 - rng.randrange(0, 4000)); end = max(p0 + 1, beg + rng.randrange(1, 15000))
        chunks = CSI.query(csi, tid, beg, end)
        assert chunks
        for j in mapped:
            t, a, b = recs[j][:3]
            if t == tid and max(a, 0) < end and max(b, 1) > beg:
                assert any(c[0] <= starts[j] and recs[j][3] <= c[1] for c in chunks), (name, tid, beg, end, j)
