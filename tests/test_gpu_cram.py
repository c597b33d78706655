"""CRAM 3.0 input through the product on the device (BamReader.cpp:482-492: the reference opens CRAM with the same reader as BAM): a handle on a CRAM gives the
counters, depth and read statistics of the BAM with the same records - for twins of the reference's BAM fixtures (tests/cram_twin.py: CRAM written by
oracle/cram_encode.py, a made-up genome) and for the reference's own CRAM fixture that carries all its bases (SampleIdentity_in_rna.cram, against the BAM that
oracle/cram_decode.py makes of it); the tools take `-in x.cram -ref genome.fa` and write what they write for the BAM."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import hostprep as H
from conftest import GOLDEN_IN as GI, ROOT

pytestmark = pytest.mark.gpu
ngsqc = __import__("importlib").import_module("ngs-bits_amd")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cram_decode as CD  # noqa: E402
import cram_encode as CE  # noqa: E402
import cram_twin  # noqa: E402

BIN = os.path.join(ROOT, "ngs-bits_amd", "bin")


@pytest.fixture(scope="module")
def twin(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("cram_twin"))
    t = cram_twin.make_twin(os.path.join(GI, "MappingQC_in2.bam"), d, max_records=20000)
    t["cram"] = os.path.join(d, "twin.cram"); CE.encode(t["bam"], t["cram"], t["genome"])
    t["cram_multi"] = os.path.join(d, "multi.cram"); CE.encode(t["bam"], t["cram_multi"], t["genome"], multi_ref=True, slice_records=900)
    t["cram_norr"] = os.path.join(d, "norr.cram"); CE.encode(t["bam"], t["cram_norr"], t["genome"], rr=False)
    # CRAM 3.1: every series in rANS Nx16 blocks of all shapes; and the file as samtools lays it out - the read names in a block of the name tokeniser,
    # which the tools never open (BamReader::skipTags / required fields)
    t["cram31"] = os.path.join(d, "v31.cram"); CE.encode(t["bam"], t["cram31"], t["genome"], version=(3, 1), methods=[50, 51, 52, 53, 54, 55, 56, 57, 58, 59])
    t["cram31_tok3"] = os.path.join(d, "v31_tok3.cram"); CE.encode(t["bam"], t["cram31_tok3"], t["genome"], version=(3, 1), methods=[51, 50, 58, 54], name_method=80)
    h = ngsqc.Handle(path=t["bam"]); h.write_bai(); h.close()      # (the tools ask for an index next to their input, as the reference does; the CRAM's .crai is written by the encoder)
    return t


def _mapping(h):
    tx, ty = H.xy_tids(h.refs)
    return h.scan_mapping(ngsqc.MODE_WGS, tid_x=tx, tid_y=ty, nonspecial=H.nonspecial(h.refs))


def _same_results(a, b):
    assert a.refs == b.refs and a.n_records == b.n_records > 1000
    ca, _ = _mapping(a); cb, _ = _mapping(b)
    assert np.array_equal(ca, cb) and int(ca.sum()) > 0
    assert np.array_equal(a.inflated(), b.inflated())                 # the same BAM stream, byte for byte
    assert np.array_equal(a.record_offsets(), b.record_offsets())
    tid = max(range(len(a.refs)), key=lambda i: a.refs[i][1])
    regs = [(tid, 1, min(a.refs[tid][1], 60000))]
    a.scan_depth(regs, min_mapq=1); b.scan_depth(regs, min_mapq=1)
    n = regs[0][2]
    assert np.array_equal(a.depth(n), b.depth(n)) and int(a.depth(n).sum()) > 0


@pytest.mark.parametrize("quals", ["device", "host"])
@pytest.mark.parametrize("which", ["cram", "cram_multi", "cram_norr", "cram31"])
def test_handle_on_a_cram_equals_the_handle_on_its_bam(twin, which, quals, monkeypatch):
    """quals: the quality arrays (rANS blocks) decoded by the kernels of csrc/cram_dev.hip into the uploaded image (the default), or on the host like the rest"""
    if quals == "host": monkeypatch.setenv("NGSQC_CRAM_DEVICE_QUALS", "0")
    else: monkeypatch.setenv("NGSQC_TIMING", "1")                                        # (the kernel time goes to stderr: profiles/r04_cram_device_quals.txt)
    ngsqc.set_reference(None if which == "cram_norr" else twin["fasta"])
    try:
        a = ngsqc.Handle(path=twin[which]); b = ngsqc.Handle(path=twin["bam"])
        try:
            _same_results(a, b)
        finally:
            a.close(); b.close()
        data = np.fromfile(twin[which], dtype=np.uint8)               # ngsqc_open_memory takes a CRAM image as well
        a = ngsqc.Handle(data=data); b = ngsqc.Handle(path=twin["bam"])
        try:
            assert a.n_records == b.n_records
            assert np.array_equal(_mapping(a)[0], _mapping(b)[0])
        finally:
            a.close(); b.close()
    finally:
        ngsqc.set_reference(None)


def test_cram_without_its_genome_is_the_references_error(twin):
    ngsqc.set_reference(None)
    with pytest.raises(ngsqc.NgsqcError) as e:
        ngsqc.Handle(path=twin["cram"])
    assert "Error while setting reference genome" in str(e.value)


def test_reference_fixture_that_carries_its_bases(tmp_path):
    """SampleIdentity_in_rna.cram (htslib-written, RR = false): the handle's BAM stream is the oracle's record for record, and the counters equal those of that BAM"""
    src = os.path.join(GI, "SampleIdentity_in_rna.cram")
    f = CD.read_cram(src); rgs = CD.read_groups(f.header)
    bam = str(tmp_path / "rna.bam")
    cram_twin.write_bam(bam, f.header, CD.ref_names(f.header), [CD.to_bam_record(r, rgs) for r in f.records])
    a = ngsqc.Handle(path=src); b = ngsqc.Handle(path=bam)
    try:
        assert a.n_records == b.n_records == len(f.records)
        assert np.array_equal(a.inflated(), b.inflated())
        assert np.array_equal(_mapping(a)[0], _mapping(b)[0])
    finally:
        a.close(); b.close()


def _run(tool, *args, env=None, ok=True):
    p = subprocess.run([os.path.join(BIN, tool)] + list(args), capture_output=True, text=True, env=dict(os.environ, **(env or {})), timeout=300)
    if ok: assert p.returncode == 0, p.stderr
    return p


def test_tools_on_a_cram_write_what_they_write_for_the_bam(twin, tmp_path):
    strip = re.compile(r"creation |<binary>|source file|twin\.(bam|cram)")
    outs = {}
    for kind in ("bam", "cram"):
        o = str(tmp_path / (kind + ".qcML"))
        # (-no_ref: no GC / AT dropout from the made-up genome; the CRAM decoder then takes the genome from NGSQC_REFERENCE)
        p = _run("MappingQC", "-in", twin[kind], "-wgs", "-build", "hg19", "-no_ref", "-out", o, env={"NGSQC_REFERENCE": twin["fasta"], "NGSQC_TIMING": "1"})
        outs[kind] = [ln for ln in open(o).read().splitlines() if not strip.search(ln)]
        if kind == "cram":   # the quality blocks went through the device decoder
            m = re.search(r"cram: (\d+) quality blocks \((\d+) bytes, (\d+) records\) decoded on the device in ([0-9.]+) ms", p.stderr)
            assert m and int(m.group(1)) >= 3 and int(m.group(3)) > 1000, p.stderr[-2000:]
            print("device quality decode:", m.group(0))
    assert outs["bam"] == outs["cram"] and len(outs["bam"]) > 30
    # a CRAM 3.1 file whose read names sit in a name-tokeniser block: the tool does not ask for names and writes the same qcML; a caller that wants every field is refused
    o = str(tmp_path / "v31.qcML")
    _run("MappingQC", "-in", twin["cram31_tok3"], "-wgs", "-build", "hg19", "-no_ref", "-out", o, env={"NGSQC_REFERENCE": twin["fasta"]})
    assert [ln for ln in open(o).read().splitlines() if not re.search(r"creation |<binary>|source file|(twin|v31_tok3)\.(bam|cram)", ln)] == outs["bam"]
    ngsqc.set_reference(twin["fasta"])
    try:
        with pytest.raises(ngsqc.NgsqcError) as e:
            ngsqc.Handle(path=twin["cram31_tok3"])
        assert e.value.code == -5 and "name tokeniser" in str(e.value)
    finally:
        ngsqc.set_reference(None)
    name, ln = max(twin["refs"], key=lambda x: x[1])
    bed = str(tmp_path / "r.bed"); open(bed, "w").write("%s\t100\t%d\n%s\t%d\t%d\n" % (name, ln // 2, name, ln // 2 + 50, ln - 10))
    cov = {k: _run("BedCoverage", "-bam", twin[k], "-in", bed, "-ref", twin["fasta"]).stdout for k in ("bam", "cram")}
    assert cov["bam"].replace("twin.bam", "X") == cov["cram"].replace("twin.cram", "X") and len(cov["bam"].splitlines()) >= 2
    low = {k: _run("BedLowCoverage", "-bam", twin[k], "-in", bed, "-cutoff", "20", "-ref", twin["fasta"]).stdout for k in ("bam", "cram")}
    assert low["bam"].replace("twin.bam", "X") == low["cram"].replace("twin.cram", "X") and len(low["bam"].splitlines()) > 3
    # without a genome: the reference's error (BamReader.cpp:486-489), exit code 1
    p = _run("BedCoverage", "-bam", twin["cram"], "-in", bed, ok=False, env={"NGSQC_REFERENCE": ""})
    assert p.returncode != 0 and "Error while setting reference genome" in (p.stderr + p.stdout)


def test_regions_on_a_cram_decode_only_their_slices(twin, tmp_path):
    """ngsqc_open_regions on a CRAM (BamReader::setRegion on a CRAM goes through the .crai): the slices whose headers overlap the regions, the depth and read counts
    of the whole file inside the region"""
    cram = str(tmp_path / "small_slices.cram"); CE.encode(twin["bam"], cram, twin["genome"], slice_records=300)
    name, ln = max(twin["refs"], key=lambda x: x[1]); tid = [n for n, _ in twin["refs"]].index(name)
    region = (name, ln // 2, ln // 2 + 200); regs = [(tid, region[1], region[2])]
    ngsqc.set_reference(twin["fasta"])
    try:
        whole = ngsqc.Handle(path=cram); part = ngsqc.Handle(path=cram, regions=[region]); head = ngsqc.Handle(path=twin["bam"])
        try:
            n = region[2] - region[1] + 1
            whole.scan_depth(regs, min_mapq=1); part.scan_depth(regs, min_mapq=1); head.scan_depth(regs, min_mapq=1)
            d = whole.depth(n)
            assert d.sum() > 0 and np.array_equal(d, part.depth(n)) and np.array_equal(d, head.depth(n))
            assert np.array_equal(whole.region_read_counts(regs, 1), part.region_read_counts(regs, 1))
            assert 0 < part.n_records < whole.n_records // 2
            with pytest.raises(ngsqc.NgsqcError):
                part.write_bai(str(tmp_path / "x.bai"))          # the index of a CRAM is a .crai
        finally:
            whole.close(); part.close(); head.close()
    finally:
        ngsqc.set_reference(None)


def test_region_names_resolve_like_the_bam_index_path(twin, tmp_path):
    """a BED that says CHR1 / 1 / chr1 for the file's chr1 must select the same CRAM slices as it selects BAM members (Chromosome::normalizedStringRepresentation:
    "chr" / "CHR" dropped, upper case) - before, only a lower-case "chr" was stripped for a CRAM and such a region silently read as depth 0"""
    cram = str(tmp_path / "small_slices.cram"); CE.encode(twin["bam"], cram, twin["genome"], slice_records=300)
    name, ln = max(twin["refs"], key=lambda x: x[1]); tid = [n for n, _ in twin["refs"]].index(name)
    bare = name[3:] if name.lower().startswith("chr") else name
    regs = [(tid, ln // 2, ln // 2 + 200)]; n = 201
    ngsqc.set_reference(twin["fasta"])
    try:
        whole = ngsqc.Handle(path=twin["bam"]); whole.scan_depth(regs, min_mapq=1); want = whole.depth(n).copy(); whole.close()
        assert want.sum() > 0
        for spelled in (name, bare, "CHR" + bare.upper(), "chr" + bare.lower()):
            for path in (cram, twin["bam"]):
                part = ngsqc.Handle(path=path, regions=[(spelled, regs[0][1], regs[0][2])])
                try:
                    part.scan_depth(regs, min_mapq=1)
                    assert np.array_equal(part.depth(n), want), (spelled, path)
                finally:
                    part.close()
    finally:
        ngsqc.set_reference(None)


def test_read_with_more_than_65535_cigar_operations(tmp_path):
    """a CRAM read of 70 002 CIGAR operations reaches the device as the BAM convention (placeholder CIGAR + CG:B,I tag), which K2 / K3 put back like
    htslib's bam_tag2cigar: the handle on the CRAM gives the inflated stream, counters and depth of the handle on the BAM the CRAM was written from"""
    d = str(tmp_path); src = os.path.join(d, "long.bam")
    cram_twin.long_cigar_bam(src)
    t = cram_twin.make_twin(src, os.path.join(d, "twin"))
    cram = os.path.join(d, "long.cram"); CE.encode(t["bam"], cram, t["genome"])
    ngsqc.set_reference(t["fasta"])
    try:
        a = ngsqc.Handle(path=cram); b = ngsqc.Handle(path=t["bam"])
        try:
            assert a.n_records == b.n_records == 141
            assert np.array_equal(a.inflated(), b.inflated())
            ca, _ = _mapping(a); cb, _ = _mapping(b)
            assert np.array_equal(ca, cb) and int(ca[9]) >= 12                       # bases_clipped: the 7S + 5S of the long read were seen through the CG tag
            regs = [(0, 1, 120000)]
            a.scan_depth(regs, min_mapq=1); b.scan_depth(regs, min_mapq=1)
            da = a.depth(120000)
            assert np.array_equal(da, b.depth(120000)) and int(da[60000]) == 1        # the long read alone covers 1 708 .. 89 207
        finally:
            a.close(); b.close()
    finally:
        ngsqc.set_reference(None)


def test_baminfo_and_readcount_on_a_cram(twin, tmp_path):
    """BamInfo names the container version (BamReader.cpp:603-617: "CRAM 3.0") and finds mapper / paired-end from the first slices; BedReadCount counts like on the BAM"""
    rows = {}
    for kind in ("bam", "cram"):
        out = _run("BamInfo", "-in", twin[kind], "-name", "-ref", twin["fasta"]).stdout.splitlines()
        assert out[0].startswith("#filename\tformat") and len(out) == 2
        rows[kind] = out[1].split("\t")
    assert rows["bam"][1] == "BAM" and rows["cram"][1] == "CRAM 3.0"
    # (the twin BAM's header spans several BGZF members and its records are cut by member boundaries - writers other than htslib do that: the head request completes
    # the record that the end of the head cuts from the members behind it)
    assert rows["bam"][2:] == rows["cram"][2:] and rows["bam"][6] == "yes" and "bwa" in rows["bam"][5]
    name, ln = max(twin["refs"], key=lambda x: x[1])
    bed = str(tmp_path / "r.bed"); open(bed, "w").write("%s\t100\t%d\n%s\t%d\t%d\n" % (name, ln // 2, name, ln // 2 + 50, ln - 10))
    cnt = {k: _run("BedReadCount", "-bam", twin[k], "-in", bed, "-ref", twin["fasta"]).stdout for k in ("bam", "cram")}
    assert cnt["bam"] == cnt["cram"] and len(cnt["bam"].splitlines()) >= 2
    assert any(int(ln_.split("\t")[-1]) > 0 for ln_ in cnt["bam"].splitlines() if not ln_.startswith("#"))
