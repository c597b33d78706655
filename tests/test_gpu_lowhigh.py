"""BedLowCoverage / BedHighCoverage and `-min_baseq` on the device against the HAND-DERIVED vectors of tests/hand_vectors.py (the
reference's own vectors need the missing panel.bam; tests/test_oracle_golden.py::test_lowhigh_hand_vectors holds the oracle to the same vectors). Two levels:
the C ABI (ngsqc_scan_depth -> per-base depth, ngsqc_lowhigh_runs) and the tools' output bytes."""
import os
import subprocess

import numpy as np
import pytest

import hand_vectors as HV
import hostprep as H
from conftest import ROOT

pytestmark = pytest.mark.gpu
ngsqc = __import__("importlib").import_module("ngs-bits_amd")
BIN = os.path.join(ROOT, "ngs-bits_amd", "bin")


@pytest.fixture(scope="module")
def hand_bam(tmp_path_factory):
    d = tmp_path_factory.mktemp("hand")
    bam = str(d / "hand.bam"); HV.write_bam(bam, HV.READS)
    h = ngsqc.Handle(path=bam)
    assert h.n_records == len(HV.READS)
    h.write_bai(); h.close()   # (random access goes through the index, like the reference's setRegion)
    return bam


@pytest.mark.parametrize("case", list(HV.CASES))
def test_depth_and_runs_through_the_c_abi(hand_bam, case, tmp_path):
    c = HV.CASES[case]
    bed = str(tmp_path / "c.bed"); HV.write_bed(bed, case)
    h = ngsqc.Handle(path=hand_bam)
    try:
        regs, _ = H.bed_regions(bed, h.refs, 2)                 # merge(true, true): what the tools scan
        h.scan_depth(regs, min_mapq=1, min_baseq=c["min_baseq"])
        assert h.depth(len(c["depth"])).tolist() == c["depth"]   # the int depth of the random-access worker; the sweep's 254 is applied where the runs are cut
        for ra in (True, False):
            runs = h.lowhigh_runs(regs, c["cutoff"], is_high=c["is_high"], saturate254=not ra)
            got = ["chr1\t%d\t%d" % (s - 1, e) for (_, s, e) in runs]
            assert got == [ln.rsplit("\t", 1)[0] for ln in HV.expected(case, ra)], (case, ra)
    finally:
        h.close()


@pytest.mark.parametrize("case", list(HV.CASES))
def test_tool_output_lines(hand_bam, case, tmp_path):
    c = HV.CASES[case]
    bed = str(tmp_path / "c.bed"); HV.write_bed(bed, case)
    tool = "BedHighCoverage" if c["is_high"] else "BedLowCoverage"
    for ra in (True, False):
        for threads in ("1", "3"):                               # the reference's tests assert the same output for every -threads
            out = str(tmp_path / f"out_{ra}_{threads}.bed")
            p = subprocess.run([os.path.join(BIN, tool), "-bam", hand_bam, "-in", bed, "-cutoff", str(c["cutoff"]), "-min_baseq", str(c["min_baseq"]), "-threads", threads, "-out", out]
                               + (["-random_access"] if ra else []), capture_output=True, text=True, timeout=300)
            assert p.returncode == 0, p.stderr
            lines = open(out).read().splitlines()
            assert [ln for ln in lines if not ln.startswith("#")] == HV.expected(case, ra), (case, ra, threads)
            if not c["is_high"]:
                n_bases = len(c["depth"]); n_regions = 1   # (every case's lines merge into one)
                assert [ln for ln in lines if ln.startswith("#")] == ["#BAM: hand.bam", "#ROI: c.bed", f"#ROI regions: {n_regions}", f"#ROI bases: {n_bases}"]


def test_depth_of_the_whole_hand_bam_in_one_scan(hand_bam, tmp_path):
    """all cases' lines in one BED: one scan, the depth arrays side by side"""
    lines = sorted({ln for c in HV.CASES.values() for ln in c["bed"]})
    bed = str(tmp_path / "all.bed"); open(bed, "w").write("".join("chr1\t%d\t%d\t%s\n" % ln for ln in lines))
    h = ngsqc.Handle(path=hand_bam)
    try:
        regs, _ = H.bed_regions(bed, h.refs, 2)
        for baseq, pick in ((0, ("c1_low", "c3_plain", "c4_plain", "c5_high", "c6_names", "c7_two_runs")), (20, ("c1_low", "c3_baseq", "c4_baseq", "c5_high", "c6_names", "c7_two_runs"))):
            h.scan_depth(regs, min_mapq=1, min_baseq=baseq)
            want = [d for k in pick for d in HV.CASES[k]["depth"]]
            assert h.depth(len(want)).tolist() == want, baseq
    finally:
        h.close()
