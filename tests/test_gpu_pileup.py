"""Site pileup on the GPU (ngsqc_site_pileup = BamReader::getPileup SNP counts for a table of sites) and the sample
contamination check built on it (Statistics::contamination): the reference's own known answers
(src/cppNGS-TEST/BamReader_Test.cpp:256-292) through the C ABI, bit-exact counts vs the oracle on every position of
windows of the fixture BAMs (CIGARs with S / N / I-only reads / deletions) and of synthetic short-read, ONT-like (CG-tag
CIGARs) and multi-tile inputs, and the MappingQC value vs the oracle's on a synthetic BAM with enough informative SNPs."""
import os
import re
import subprocess

import numpy as np
import pytest

import bamgen_lib as G
import hostprep as H
import oracle_lib as O
from conftest import GOLDEN_IN as GI, RESOURCES, ROOT

pytestmark = pytest.mark.gpu
ngsqc = __import__("importlib").import_module("ngs-bits_amd")


def _tid(h, chrom):
    return H.tid_map(h.refs)[H.chr_num(chrom)]


def test_reference_known_answers_through_the_c_abi():
    h = ngsqc.Handle(path=os.path.join(GI, "BamReader_rna.bam"))
    t10, t11 = _tid(h, "chr10"), _tid(h, "chr11")
    c = h.site_pileup([(t10, 90974727), (t10, 92675287), (t11, 92675295)])
    assert c[0, [0, 1, 2, 3, 5]].sum() == 132 and abs(c[0, 1] / (c[0, 0] + c[0, 1]) - 0.4621) < 0.001   # BamReader_Test.cpp:261-264
    assert c[1, [0, 1, 2, 3, 5]].sum() == 23 and c[2].sum() == 0                                          # :266-274
    h.close()
    h = ngsqc.Handle(path=os.path.join(GI, "BamReader_insert_only.bam"))
    t = _tid(h, "chr19")
    c = h.site_pileup([(t, 5787214), (t, 5787215)])
    assert c[0, [0, 1, 2, 3, 5]].sum() == 111 and abs(c[0, 1] / (c[0, 3] + c[0, 1]) - 0.556) < 0.001     # :284-286
    assert c[1, [0, 1, 2, 3, 5]].sum() == 118 and abs(c[1, 0] / (c[1, 2] + c[1, 0]) - 0.389) < 0.001     # :289-291
    h.close()


def _compare(path, sites, **kw):
    ob = O.Bam(path)
    h = ngsqc.Handle(path=path)
    try:
        got = h.site_pileup(sites, **kw)
        exp = O.site_pileup(ob, sites, **kw)
        assert (got[:, 6:] == 0).all()
        bad = np.nonzero((got[:, :6] != exp).any(axis=1))[0]
        assert bad.size == 0, (sites[int(bad[0])], got[int(bad[0])], exp[int(bad[0])], kw)
        return int(exp.sum())
    finally:
        h.close()


@pytest.mark.parametrize("bam,chrom,center", [("BamReader_rna.bam", "chr10", 90974727), ("BamReader_rna.bam", "chr10", 92675287),
                                              ("BamReader_insert_only.bam", "chr19", 5787214), ("MappingQC_in2.bam", None, None),
                                              ("BamReader_lr.bam", None, None), ("Statistics_longread.bam", None, None)])
def test_every_position_of_a_window_matches_the_oracle(bam, chrom, center):
    path = os.path.join(GI, bam)
    ob = O.Bam(path)
    if chrom is None:   # window around the first mapped record
        offs = ob.record_offsets(); raw = ob.inflated()
        k = min(10, len(offs) - 1)
        tid, pos0 = (int(x) for x in np.frombuffer(raw[int(offs[k]) + 4:int(offs[k]) + 12].tobytes(), dtype="<i4"))
        center = pos0 + 500
    else:
        tid = H.tid_map(ob.refs)[H.chr_num(chrom)]
    sites = [(tid, p) for p in range(center - 400, center + 401)]
    total = 0
    for kw in (dict(), dict(min_baseq=0), dict(include_not_properly_paired=True), dict(min_mapq=20, min_baseq=30, include_not_properly_paired=True)):
        total += _compare(path, sites, **kw)
    assert total > 0


def test_synthetic_short_long_and_tiled(tmp_path, monkeypatch):
    rng = np.random.default_rng(7)
    for name, gen in (("sr.bam", dict(n_reads=120_000, seed=41, start_pos=15_900_000)),
                      ("sr_unaligned.bam", dict(n_reads=60_000, seed=42, aligned=False, start_pos=15_900_000)),
                      ("ont.bam", dict(n_reads=900, seed=43, mode=1, depth=40.0, start_pos=15_900_000))):
        path = str(tmp_path / name)
        G.write(path, **gen)
        ob = O.Bam(path)
        tid = H.tid_map(ob.refs)[1]
        sites = sorted({(tid, int(p)) for p in rng.integers(15_900_000, 16_700_000, 6000)})
        assert _compare(path, sites, include_not_properly_paired=True) > 1000
        _compare(path, sites, min_baseq=20)
    monkeypatch.setenv("NGSQC_TILE_MEMBERS", "3")
    assert _compare(str(tmp_path / "sr_unaligned.bam"), sites, include_not_properly_paired=True) > 1000


def _known_snvs(build, refs):
    """NGSHelper::getKnownVariants(build, true, 0.2, 0.8) on the resource table, as (tid, pos, ref, alt)."""
    tm = H.tid_map(refs); out = []
    for ln in open(os.path.join(RESOURCES, f"{build}_snps.tsv")):
        c, p, r, a, af = ln.rstrip("\n").split("\t")
        try:
            f = float(af)
        except ValueError:
            f = 0.0
        a0 = a.split(",")[0].upper()
        if 0.2 <= f <= 0.8 and len(r) == 1 and len(a0) == 1 and a0 != "-" and r != "-":
            out.append((tm[H.chr_num(c)], int(p), r, a0))
    return out


def test_contamination_value_of_the_tool_matches_the_oracle(tmp_path):
    path = str(tmp_path / "cont.bam")
    G.write(path, n_reads=560_000, seed=44, first_contig=5, start_pos=32_200_000, depth=80.0)   # chr6:32.2-33.2 Mb holds 264 of the known SNVs
    open(path + ".bai", "wb").close()                       # getPileup needs an index in the reference; the GPU path only checks that it exists
    ob = O.Bam(path)
    want = O.contamination(ob, _known_snvs("hg38", ob.refs))
    assert want != "n/a"                                     # enough informative SNPs: the numeric branch is exercised
    out = str(tmp_path / "cont.qcML")
    p = subprocess.run([os.path.join(ROOT, "ngs-bits_amd", "bin", "MappingQC"), "-in", path, "-wgs", "-build", "hg38", "-no_ref", "-out", out], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    m = re.search(r'name="SNV allele frequency deviation"[^>]*value="([^"]*)"', open(out).read())
    assert m and m.group(1) == want, (m and m.group(1), want)
    # -no_cont leaves the value out
    p = subprocess.run([os.path.join(ROOT, "ngs-bits_amd", "bin", "MappingQC"), "-in", path, "-wgs", "-build", "hg38", "-no_ref", "-no_cont", "-out", out], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "SNV allele frequency deviation" not in open(out).read()


@pytest.mark.parametrize("kind", ["short", "short_tiles", "ont", "unaligned"])
def test_pileup_over_the_riding_scans_candidates_equals_the_full_pass(tmp_path, monkeypatch, kind):
    """Round 4: in a fused job the scan that rides K2's chain walk names the records whose reference span holds a known site (deferred long-CIGAR records: all that pass
    the read filters), and the site pileup reads only those. Same site counts as the pass over every record (NGSQC_NO_FUSED_PILEUP=1), as the stand-alone pileup and as
    the oracle - short reads (one tile and tiny tiles), ONT-like reads with CG-tag CIGARs (every record deferred), unaligned members (the riding scan is taken back)."""
    path = str(tmp_path / "p.bam")
    gen = {"short": dict(n_reads=150_000, seed=51), "short_tiles": dict(n_reads=150_000, seed=52), "ont": dict(n_reads=1200, seed=53, mode=1, depth=40.0),
           "unaligned": dict(n_reads=60_000, seed=54, aligned=False)}[kind]
    G.write(path, start_pos=15_900_000, **gen)
    if kind == "short_tiles":
        monkeypatch.setenv("NGSQC_TILE_MEMBERS", "40")
    ob = O.Bam(path)
    h = ngsqc.Handle(path=path)
    regs, _ = H.bed_regions(os.path.join(RESOURCES, "hg38_440_omim_genes.bed"), h.refs, 3); tx, ty = H.xy_tids(h.refs)
    t1 = _tid(h, "chr1")
    lo = 15_900_000 + 2_000; span = 600_000 if kind != "ont" else 1_500_000
    rng = np.random.default_rng(7)
    sites = sorted({(t1, int(p)) for p in rng.integers(lo, lo + span, 400)})
    mp = dict(mode=ngsqc.MODE_WGS, regions=regs, min_mapq=1, tid_x=tx, tid_y=ty, nonspecial=H.nonspecial(h.refs))
    a = h.run_job(mapping=mp, sites=sites, site_params=(1, 13, kind == "ont"))
    h.drop_decoded()
    monkeypatch.setenv("NGSQC_NO_FUSED_PILEUP", "1")
    b = h.run_job(mapping=mp, sites=sites, site_params=(1, 13, kind == "ont"))
    monkeypatch.delenv("NGSQC_NO_FUSED_PILEUP")
    c = h.site_pileup(sites, min_mapq=1, min_baseq=13, include_not_properly_paired=(kind == "ont"))
    h.close()
    assert np.array_equal(a["site_counts"], b["site_counts"]) and np.array_equal(a["site_counts"], c) and np.array_equal(a["counters"], b["counters"])
    assert int(np.asarray(a["site_counts"])[:, :6].sum()) > 1000
    exp = O.site_pileup(ob, sites, 1, 13, kind == "ont")   # int64[n, 6]: A, C, G, T, N, deletion
    assert np.array_equal(np.asarray(a["site_counts"])[:len(sites), :6], exp)
