"""Pins the CPU restatement (oracle/) against the reference's own known-answer tests and expected tool outputs.

Constants are the ones in /root/reference/src/cppNGS-TEST/Statistics_Test.cpp (line numbers cited per test) and the
genome-independent lines of /root/reference/src/tools-TEST/data_out/MappingQC_test*_out.* (copied as data fixtures to
tests/golden/ref_out). Lines that need an hg19/hg38 FASTA (AT/GC dropout) are excluded: no genome exists in the image.
"""
import os
import re

import numpy as np
import pytest

import oracle_lib as O
from conftest import GOLDEN_IN as GI, GOLDEN_OUT as GO, RESOURCES

GENOME_DEPENDENT = {"AT dropout", "GC dropout"}


def _bam(name):
    return O.Bam(os.path.join(GI, name))


def _check_against_expected(res, expected_file):
    exp = open(os.path.join(GO, expected_file)).read()
    checked = 0
    for name, value in res.values().items():
        if name in GENOME_DEPENDENT:
            continue
        if expected_file.endswith(".txt"):
            assert f"{name}: {value}\n" in exp, (name, value)
        else:
            pat = r'name="%s" description="[^"]*" value="%s"' % (re.escape(name), re.escape(value))
            assert re.search(pat, exp), (name, value)
        checked += 1
    return checked


def test_mapping_close_exons():  # Statistics_Test.cpp:300-362
    r = O.mapping(_bam("close_exons.bam"), O.MODE_ROI, os.path.join(GI, "close_exons.bed"))
    v = r.values()
    exp = {"trimmed base percentage": "20.88", "clipped base percentage": "0.31", "mapped read percentage": "99.93",
           "on-target read percentage": "99.86", "near-target read percentage": "99.93",
           "properly-paired read percentage": "97.37", "insert size": "116.95",
           "duplicate read percentage": "n/a (no duplicates marked or duplicates removed during data analysis)",
           "bases usable (MB)": "0.06", "target region read depth": "388.79", "target region 10x percentage": "100.00",
           "target region 20x percentage": "100.00", "target region 30x percentage": "100.00",
           "target region 50x percentage": "100.00", "target region 60x percentage": "100.00",
           "target region 100x percentage": "93.51", "target region 200x percentage": "79.87",
           "target region 500x percentage": "30.52", "target region half depth percentage": "79.87"}
    for k, e in exp.items():
        assert v[k] == e, k
    assert abs(r["bases_usable_no_overlap"] / r["roi_bases"] - 200.448) < 1e-3
    assert len(r.lines) == 25  # I_EQUAL(stats.count(), 25)
    assert r["roi_bases"] == 154 and r.depth.size == 154 and int(r.depth.sum()) == r["bases_usable"]


def test_mapping_noroi_and_wgs_without_roi():  # Statistics_Test.cpp:364-425
    b = _bam("close_exons.bam")
    for mode in (O.MODE_NOROI, O.MODE_WGS):
        r = O.mapping(b, mode)
        v = r.values()
        assert v["trimmed base percentage"] == "20.88" and v["clipped base percentage"] == "0.31"
        assert v["mapped read percentage"] == "99.93" and v["on-target read percentage"] == "99.93"
        assert v["properly-paired read percentage"] == "97.37" and v["insert size"] == "116.95"
        assert v["bases usable (MB)"] == "0.17"
        assert len(r.lines) == 11


def test_mapping_wgs_with_roi():  # Statistics_Test.cpp:427-478
    r = O.mapping(_bam("Statistics_mapqc_wgs.bam"), O.MODE_WGS, os.path.join(GI, "Statistics_mapqc_wgs.bed"), merge_bed=False)
    v = r.values()
    exp = {"trimmed base percentage": "0.16", "clipped base percentage": "0.62", "mapped read percentage": "99.77",
           "on-target read percentage": "99.77", "properly-paired read percentage": "98.60", "insert size": "419.29",
           "duplicate read percentage": "0.77", "bases usable (MB)": "0.33", "target region 10x percentage": "22.10",
           "target region 20x percentage": "14.94", "target region 30x percentage": "10.80",
           "target region 50x percentage": "6.41", "target region 60x percentage": "4.77",
           "target region 100x percentage": "1.24", "target region 200x percentage": "0.00",
           "target region 500x percentage": "0.00", "target region half depth percentage": "26.99"}
    for k, e in exp.items():
        assert v[k] == e, k
    assert len(r.lines) == 24


def test_mapping_cfdna():  # Statistics_Test.cpp:481-566 == tools-TEST MappingQC_test08
    r = O.mapping(_bam("MappingQC_in4.bam"), O.MODE_ROI, os.path.join(GI, "cfDNA.bed"), cfdna=True)
    v = r.values()
    assert v["target region read depth"] == "6505.90"
    assert v["target region read depth 2-fold duplication"] == "4668.92"
    assert v["target region read depth 3-fold duplication"] == "4568.45"
    assert v["target region read depth 4-fold duplication"] == "4531.60"
    assert v["raw target region read depth"] == "242405.61"
    assert v["target region 5000x percentage"] == "93.75" and v["target region 7500x percentage"] == "46.25"
    assert abs(r["bases_usable_no_overlap"] / r["roi_bases"] - 4936.42) < 0.01
    assert len(r.lines) == 37


@pytest.mark.parametrize("bam,mode,bed,merge,cfdna,expected,nmin", [
    ("MappingQC_in2.bam", O.MODE_ROI, "MappingQC_in2.bed", True, False, "MappingQC_test02_out.txt", 20),
    ("MappingQC_in1.bam", O.MODE_ROI, "MappingQC_in2.bed", True, False, "MappingQC_test03_out.qcML", 19),
    ("MappingQC_in2.bam", O.MODE_WGS, "@hg19_439_omim_genes.bed", False, False, "MappingQC_test04_out.qcML", 19),
    ("MappingQC_in1.bam", O.MODE_WGS, "@hg19_439_omim_genes.bed", False, False, "MappingQC_test05_out.qcML", 18),
    ("MappingQC_in3.bam", O.MODE_NOROI, None, False, False, "MappingQC_test07_out.qcML", 10),
    ("MappingQC_in4.bam", O.MODE_ROI, "MappingQC_in3.bed", True, True, "MappingQC_test08_out.qcML", 30),
    ("MappingQC_in5.bam", O.MODE_WGS, "@hg38_440_omim_genes.bed", False, False, "MappingQC_test10_out.qcML", 19),
])
def test_tool_expected_outputs(bam, mode, bed, merge, cfdna, expected, nmin):
    """src/tools-TEST/MappingQC_Test.cpp:30-117 (test02/03/04/05/07/08/10), genome-independent lines."""
    if bed and bed.startswith("@"):
        bed = os.path.join(RESOURCES, bed[1:])
    elif bed:
        bed = os.path.join(GI, bed)
    r = O.mapping(_bam(bam), mode, bed, merge_bed=merge, cfdna=cfdna)
    assert _check_against_expected(r, expected) >= nmin


def test_avg_coverage_1decimal():  # Statistics_Test.cpp:759-776
    for ra in (False, True):
        cov, text, _ = O.avg_coverage(_bam("close_exons.bam"), os.path.join(GI, "close_exons.bed"), merge_bed=True, min_mapq=20,
                                      decimals=1, random_access=ra)
        assert text.splitlines() == ["chr1\t45332752\t45332844\t454.0", "chr1\t45332907\t45332969\t292.1"]


def test_low_coverage_known_answers(tmp_path):  # Statistics_Test.cpp:691-712
    b = _bam("close_exons.bam")
    for ra in (True, False):
        r = O.low_high_coverage(b, os.path.join(GI, "close_exons.bed"), 20, 1, tool_merge=2, random_access=ra)
        assert r["roi_bases"] == 154 and r["out_bases"] == 0
    bed = tmp_path / "r.bed"
    bed.write_text("chr13\t32931868\t32931970\n")
    for f in ("lowcov_bug_case1.bam", "lowcov_bug_case2.bam"):
        for ra in (True, False):
            assert O.low_high_coverage(_bam(f), str(bed), 20, 1, tool_merge=0, random_access=ra)["out_bases"] == 0


def test_lowhigh_hand_vectors(tmp_path):
    """rows a10 / a11 (BedLowCoverage / BedHighCoverage, `-min_baseq`): every reference vector needs the missing panel.bam, so the oracle is pinned on vectors
    written out BY HAND from the reference's lines (tests/hand_vectors.py: the '=' / 'X' quirk of BamAlignment::qualities, D / N stay covered, I / S advance the
    read index, 300 -> 254 in the sweep only, names joined by the input merge and made unique by the output merge, two runs in one line)"""
    import hand_vectors as HV
    bam = str(tmp_path / "hand.bam"); HV.write_bam(bam, HV.READS)
    ob = O.Bam(bam)
    assert ob.count == len(HV.READS) == 312
    for case, c in HV.CASES.items():
        bed = str(tmp_path / (case + ".bed")); HV.write_bed(bed, case)
        for ra in (True, False):
            r = O.low_high_coverage(ob, bed, c["cutoff"], 1, c["min_baseq"], is_high=c["is_high"], random_access=ra, tool_merge=1)
            assert r["bed"].splitlines() == HV.expected(case, ra), (case, ra)
            assert r["depth"].tolist() == (c["depth"] if ra else [min(d, 254) for d in c["depth"]]), (case, ra)   # (the sweep's unsigned char stops at 254)
            assert r["roi_bases"] == len(c["depth"])
            r = O.low_high_coverage(ob, bed, c["cutoff"], 1, c["min_baseq"], is_high=c["is_high"], random_access=ra, tool_merge=0)
            assert r["bed"].splitlines() == HV.expected(case, ra, function_level=True), (case, ra, "function level")


def test_yx_longread_and_sry(tmp_path):  # Statistics_Test.cpp:811-820, 850-854
    r = O.mapping(_bam("Statistics_longread.bam"), O.MODE_NOROI)
    assert r["reads_x"] == 214 and r["reads_y"] == 0
    bed = tmp_path / "sry.bed"
    bed.write_text("chrY\t2655030\t2655641\n")
    _, text, _ = O.avg_coverage(_bam("sry.bam"), str(bed), min_mapq=1, decimals=2)
    assert text.strip().endswith("\t67.27")


def test_bam_reader_accessors():  # BamReader_Test.cpp: CIGAR with thousands of ops / CG tag handling on long reads
    b = _bam("BamReader_lr.bam")
    assert b.count > 0 and b.n_blocks > 0
    offs = b.record_offsets()
    assert offs[0] == b.first_record_offset and np.all(np.diff(offs) > 36)


def _pile(ob, chrom, pos, **kw):
    import hostprep as H
    refs = ob.refs
    tid = H.tid_map(refs)[H.chr_num(chrom)]
    a, c, g, t, n, d = (int(x) for x in O.site_pileup(ob, [(tid, pos)], **kw)[0])
    return dict(A=a, C=c, G=g, T=t, N=n, d=d, depth=a + c + g + t, depth_del=a + c + g + t + d)


def test_pileup_known_answers_rna():  # BamReader_Test.cpp:256-275 (CIGARs with S and N operations)
    ob = O.Bam(os.path.join(GI, "BamReader_rna.bam"))
    p = _pile(ob, "chr10", 90974727)
    assert p["depth_del"] == 132 and abs(p["C"] / (p["A"] + p["C"]) - 0.4621) < 0.001
    assert _pile(ob, "chr10", 92675287)["depth_del"] == 23
    assert _pile(ob, "chr11", 92675295)["depth_del"] == 0


def test_pileup_known_answers_insert_only():  # BamReader_Test.cpp:278-292 (reads whose CIGAR is insertions / soft-clips only)
    ob = O.Bam(os.path.join(GI, "BamReader_insert_only.bam"))
    p = _pile(ob, "chr19", 5787214)
    assert p["depth_del"] == 111 and abs(p["C"] / (p["T"] + p["C"]) - 0.556) < 0.001
    p = _pile(ob, "chr19", 5787215)
    assert p["depth_del"] == 118 and abs(p["A"] / (p["G"] + p["A"]) - 0.389) < 0.001


def test_reads_qc_known_answers():  # src/tools-TEST/MappingQC_Test.cpp:78-91 -> data_out/MappingQC_test11_out.qcML (read_qc of MappingQC_in5.bam)
    q = O.reads_qc(O.Bam(os.path.join(GI, "MappingQC_in5.bam")))
    exp = dict(re.findall(r'name="([^"]+)" description="[^"]*" value="([^"]*)"', open(os.path.join(GO, "MappingQC_test11_out.qcML"), encoding="latin-1").read()))
    total = q["c_forward"] + q["c_reverse"]; bases_total = int(q["bases"].sum()); lens = np.nonzero(q["read_lengths"])[0]
    assert str(total) == exp["read count"]
    assert (f"{lens[0]}-{lens[-1]}" if lens.size >= 4 else ", ".join(str(x) for x in lens)) == exp["read length"]
    assert f"{q['bases_sequenced'] / 1e6:.2f}" == exp["bases sequenced (MB)"]
    assert f"{100.0 * q['c_read_q20'] / total:.2f}" == exp["Q20 read percentage"]
    assert f"{100.0 * q['c_base_q20'] / bases_total:.2f}" == exp["Q20 base percentage"]
    assert f"{100.0 * q['c_base_q30'] / bases_total:.2f}" == exp["Q30 base percentage"]
    assert f"{100.0 * q['bases'][4] / bases_total:.2f}" == exp["no base call percentage"]
    assert f"{100.0 * (q['bases'][1] + q['bases'][2]) / (bases_total - q['bases'][4]):.2f}" == exp["gc content percentage"]
    # internal consistency of the derived counters the GPU side does not transfer
    assert q["c_base_q20"] == q["base_qualities"][20:].sum() and q["c_base_q30"] == q["base_qualities"][30:].sum()
    assert q["c_read_q20"] == q["qscore_dist_r1"][20:].sum() + q["qscore_dist_r2"][20:].sum()


def test_header_only_bam_prints_nan_like_qt(tmp_path):
    """no reads at all: the reference's percentages are 0 / 0, and QCValue::toString -> QString::number(NaN, 'f', 2) prints "nan" (QCCollection.cpp:121-126) where printf would
    print "-nan"; everything that does not divide by the read count stays a number (an empty ROI depth array is 100 % covered at half depth 0)"""
    import hand_vectors as HV
    bam = str(tmp_path / "empty.bam"); HV.write_bam(bam, [])
    bed = str(tmp_path / "roi.bed"); open(bed, "w").write("chr1\t100\t200\tx\n")
    ob = O.Bam(bam)
    assert ob.count == 0
    v = O.mapping(ob, O.MODE_ROI, bed, merge_bed=True).values()
    assert v["mapped read percentage"] == v["on-target read percentage"] == v["trimmed base percentage"] == v["clipped base percentage"] == "nan"
    assert v["target region read depth"] == "0.00" and v["target region 20x percentage"] == "0.00" and v["target region half depth percentage"] == "100.00"
    assert O.mapping(ob, O.MODE_NOROI).values()["mapped read percentage"] == "nan"
    assert O.avg_coverage(ob, bed)[1] == "chr1\t100\t200\tx\t0.00\n"
    assert O.low_high_coverage(ob, bed, 5)["bed"].splitlines() == ["chr1\t100\t200\tx"]
