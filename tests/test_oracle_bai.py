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
