"""CRAM 3.0 input on the host (csrc/cram.hip through ngsqc_cram_to_bam, no GPU): the BAM records the product builds from a CRAM equal, byte for byte, the records
oracle/cram_decode.py builds (pinned on the reference's fixtures and known answers: tests/test_oracle_cram.py) - for the reference's CRAM fixtures (those that need
the hg38 genome in the mode without a genome: reference-derived bases are N on both sides) and for CRAM files written by oracle/cram_encode.py from the
reference's BAM fixtures with a made-up genome (there the decoded records must equal the BAM's own)."""
import os
import struct
import sys
import zlib

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GI = os.path.join(HERE, "golden", "ref_in")
ngsqc = __import__("importlib").import_module("ngs-bits_amd")
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import cram_decode as CD  # noqa: E402


def bam_stream(path):
    img = open(path, "rb").read(); pos = 0; out = bytearray(); stored = 0; n = 0
    while pos < len(img):
        bs = struct.unpack_from("<H", img, pos + 16)[0] + 1
        if img[pos + 18] == 1: stored += 1
        raw = zlib.decompress(img[pos + 18:pos + bs - 8], -15)
        assert zlib.crc32(raw) == struct.unpack_from("<I", img, pos + bs - 8)[0]
        out += raw; pos += bs; n += 1
    return bytes(out), stored, n


def split_bam(stream):
    assert stream[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", stream, 4)[0]; text = stream[8:8 + l_text]; o = 8 + l_text
    n_ref = struct.unpack_from("<i", stream, o)[0]; o += 4; refs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", stream, o)[0]; refs.append((stream[o + 4:o + 4 + ln - 1].decode(), struct.unpack_from("<i", stream, o + 4 + ln)[0])); o += 8 + ln
    recs = []
    while o < len(stream):
        bs = struct.unpack_from("<i", stream, o)[0]; recs.append(stream[o:o + 4 + bs]); o += 4 + bs
    return text, refs, recs


@pytest.mark.parametrize("name", ["cramTest.cram", "SampleIdentity_in_wes.cram", "SampleIdentity_in_rna.cram"])
def test_fixture_records_equal_the_oracle(name, tmp_path, monkeypatch):
    src = os.path.join(GI, name); out = str(tmp_path / "t.bam")
    f = CD.read_cram(src)
    needs_genome = any(ch.RR for _, _, ch in f.containers)
    if needs_genome:
        ngsqc.set_reference(None); monkeypatch.delenv("NGSQC_REFERENCE", raising=False)
        with pytest.raises(ngsqc.NgsqcError) as e:    # BamReader.cpp:486-489
            ngsqc.cram_to_bam(src, out)
        assert "Error while setting reference genome" in str(e.value)
        monkeypatch.setenv("NGSQC_CRAM_NO_REFERENCE", "1")
    ngsqc.cram_to_bam(src, out)
    stream, stored, n_members = bam_stream(out)
    assert stored == n_members - 1                      # stored deflate blocks and the EOF member
    text, refs, recs = split_bam(stream)
    assert text.decode() == f.header and refs == CD.ref_names(f.header)
    rgs = CD.read_groups(f.header)
    assert len(recs) == len(f.records)
    for i, (got, r) in enumerate(zip(recs, f.records)):
        want = CD.to_bam_record(r, rgs)
        assert got == want, (i, r.name, got[:60].hex(), want[:60].hex())


def test_damaged_and_unsupported_files(tmp_path, monkeypatch):
    monkeypatch.setenv("NGSQC_CRAM_NO_REFERENCE", "1")
    d = bytearray(open(os.path.join(GI, "SampleIdentity_in_rna.cram"), "rb").read()); out = str(tmp_path / "o.bam")
    for at in (len(d) // 2, 2000, len(d) - 100):
        bad = bytearray(d); bad[at] ^= 0x41
        p = str(tmp_path / "bad.cram"); open(p, "wb").write(bad)
        with pytest.raises(ngsqc.NgsqcError) as e:     # every block and container header carries a CRC-32
            ngsqc.cram_to_bam(p, out)
        assert "Could not read next alignment in BAM/CRAM file" in str(e.value)
    p = str(tmp_path / "cut.cram"); open(p, "wb").write(d[:len(d) // 3])
    with pytest.raises(ngsqc.NgsqcError):
        ngsqc.cram_to_bam(p, out)
    v32 = bytearray(d); v32[5] = 2
    p = str(tmp_path / "v32.cram"); open(p, "wb").write(v32)
    with pytest.raises(ngsqc.NgsqcError) as e:
        ngsqc.cram_to_bam(p, out)
    assert e.value.code == -5 and "CRAM 3.2" in str(e.value)          # NGSQC_E_UNSUPPORTED
    with pytest.raises(ngsqc.NgsqcError):
        ngsqc.cram_to_bam(os.path.join(GI, "sry.bam"), out)


# ---- BAM truth: CRAM files written by oracle/cram_encode.py from the reference's BAM fixtures (moved onto short contigs with a made-up genome: tests/cram_twin.py) ----
import cram_encode as CE  # noqa: E402
import cram_twin  # noqa: E402
from test_oracle_cram import VARIANTS  # noqa: E402


@pytest.fixture(scope="module")
def twins(tmp_path_factory):
    out = {}
    for src in ("MappingQC_in2.bam", "BamReader_rna.bam", "MappingQC_in5.bam"):
        d = str(tmp_path_factory.mktemp("twin_" + src[:-4]))
        out[src] = cram_twin.make_twin(os.path.join(GI, src), d, max_records=6000)
    return out


@pytest.mark.parametrize("src", ["MappingQC_in2.bam", "BamReader_rna.bam", "MappingQC_in5.bam"])
@pytest.mark.parametrize("variant", list(VARIANTS))
def test_cram_of_a_bam_gives_back_the_bam(twins, src, variant, tmp_path, monkeypatch):
    twin = twins[src]; cram = str(tmp_path / "twin.cram"); out = str(tmp_path / "back.bam")
    CE.encode(twin["bam"], cram, twin["genome"], **VARIANTS[variant])
    monkeypatch.delenv("NGSQC_CRAM_NO_REFERENCE", raising=False)
    if variant in ("no_genome_needed", "embedded_reference"):
        ngsqc.set_reference(None)                      # every base is in the file
    else:
        ngsqc.set_reference(twin["fasta"])
    try:
        ngsqc.cram_to_bam(cram, out)
    finally:
        ngsqc.set_reference(None)
    text, refs, recs = split_bam(bam_stream(out)[0])
    assert text.decode() == twin["text"] and refs == twin["refs"]
    assert len(recs) == len(twin["records"]) > 500
    for i, (got, want) in enumerate(zip(recs, twin["records"])):
        assert got == want, (i, got[:48].hex(), want[:48].hex())


def test_read_with_more_than_65535_cigar_operations(tmp_path, monkeypatch):
    """n_cigar_op is 16 bits. A CRAM read of 70 002 operations comes back as the BAM convention (placeholder <l_seq>S<ref_len>N + CG:B,I behind the
    other tags) - the record of the BAM the CRAM was written from, byte for byte, from the product and from the oracle."""
    d = str(tmp_path); src = os.path.join(d, "long.bam")
    cram_twin.long_cigar_bam(src)
    twin = cram_twin.make_twin(src, os.path.join(d, "twin"))
    long_raw = [r for r in twin["records"] if b"long_cigar\0" in r[:64]]
    assert len(long_raw) == 1 and struct.unpack_from("<H", long_raw[0], 16)[0] == 2 and b"CGBI" in long_raw[0]
    parsed = CE.parse_record(long_raw[0])
    assert len(parsed["cigar"]) == 70002 and all(t[0] != b"CG" for t in parsed["tags"])          # the encoder sees the real operations, like htslib's reader
    cram = os.path.join(d, "long.cram"); out = os.path.join(d, "back.bam")
    CE.encode(twin["bam"], cram, twin["genome"])
    monkeypatch.delenv("NGSQC_CRAM_NO_REFERENCE", raising=False)
    ngsqc.set_reference(twin["fasta"])
    try:
        ngsqc.cram_to_bam(cram, out)
    finally:
        ngsqc.set_reference(None)
    _, _, recs = split_bam(bam_stream(out)[0])
    assert recs == twin["records"]
    f = CD.read_cram(cram, cram_twin.ref_fetch_of(twin)); rgs = CD.read_groups(f.header)
    assert [CD.to_bam_record(r, rgs) for r in f.records] == twin["records"]


@pytest.mark.parametrize("name", ["cramTest.cram", "SampleIdentity_in_wes.cram", "twin"])
def test_names_and_tags_are_not_decoded_when_nobody_needs_them(name, twins, tmp_path, monkeypatch):
    """ngsqc_set_cram_skip (BamReader::skipTags / CRAM_OPT_REQUIRED_FIELDS in the reference, BamReader.cpp:525-572): the external blocks of the read names and of
    the optional fields are not inflated; every record is the oracle's record with the name "*" and no tags but RG - every other byte (positions, CIGAR, bases,
    qualities, mate fields) as before."""
    out = str(tmp_path / "s.bam"); full = str(tmp_path / "f.bam")
    if name == "twin":
        twin = twins["MappingQC_in2.bam"]; src = str(tmp_path / "t.cram"); CE.encode(twin["bam"], src, twin["genome"])
        fetch = cram_twin.ref_fetch_of(twin); ngsqc.set_reference(twin["fasta"]); monkeypatch.delenv("NGSQC_CRAM_NO_REFERENCE", raising=False)
    else:
        src = os.path.join(GI, name); fetch = None; ngsqc.set_reference(None); monkeypatch.setenv("NGSQC_CRAM_NO_REFERENCE", "1")
    try:
        ngsqc.cram_to_bam(src, full)
        for flags in (ngsqc.CRAM_SKIP_NAMES | ngsqc.CRAM_SKIP_TAGS, ngsqc.CRAM_SKIP_NAMES, ngsqc.CRAM_SKIP_TAGS):
            ngsqc.set_cram_skip(flags)
            try:
                ngsqc.cram_to_bam(src, out)
            finally:
                ngsqc.set_cram_skip(0)
            f = CD.read_cram(src, fetch); rgs = CD.read_groups(f.header)
            _, _, recs = split_bam(bam_stream(out)[0])
            assert len(recs) == len(f.records) > 1000
            for i, (got, r) in enumerate(zip(recs, f.records)):
                if flags & ngsqc.CRAM_SKIP_NAMES: r.name = None
                if flags & ngsqc.CRAM_SKIP_TAGS: r.tags = []
                want = CD.to_bam_record(r, rgs)
                assert got == want, (flags, i, got[:60].hex(), want[:60].hex())
            assert bam_stream(out)[0] != bam_stream(full)[0]
    finally:
        ngsqc.set_reference(None)


def test_cram31_name_tokeniser_block_is_refused_only_when_names_are_wanted(twins, tmp_path, monkeypatch):
    """CRAM 3.1 as samtools writes it keeps the read names in a block of the name tokeniser (method 8), which this build does not decode (nor the adaptive arithmetic
    coder, 6, nor fqzcomp, 7): NGSQC_E_UNSUPPORTED - but only for a caller that asks for names. The tools never do (BamReader::skipTags / required fields,
    BamReader.cpp:525-572): for them the block's CRC is checked and its bytes are not looked at, every other series comes out of rANS Nx16 blocks."""
    twin = twins["MappingQC_in2.bam"]; src = str(tmp_path / "t31.cram"); out = str(tmp_path / "o.bam")
    CE.encode(twin["bam"], src, twin["genome"], version=(3, 1), methods=[50, 51, 54, 55, 58], name_method=80)
    ngsqc.set_reference(twin["fasta"]); monkeypatch.delenv("NGSQC_CRAM_NO_REFERENCE", raising=False)
    try:
        with pytest.raises(ngsqc.NgsqcError) as e:
            ngsqc.cram_to_bam(src, out)
        assert e.value.code == -5 and "name tokeniser" in str(e.value)
        ngsqc.set_cram_skip(ngsqc.CRAM_SKIP_NAMES)
        try:
            ngsqc.cram_to_bam(src, out)
        finally:
            ngsqc.set_cram_skip(0)
        _, _, recs = split_bam(bam_stream(out)[0])
        assert len(recs) == len(twin["records"]) > 1000
        for got, want in zip(recs, twin["records"]):
            l_name = want[12]; w = bytearray(want[:36]) + b"*\0" + want[36 + l_name:]   # the BAM's record with the name "*"
            w[12] = 2; struct.pack_into("<i", w, 0, len(w) - 4)
            assert got == bytes(w)
    finally:
        ngsqc.set_reference(None)


def test_read_names_that_share_a_block_with_a_series_that_stays(twins, tmp_path, monkeypatch):
    """RN, IN (inserted bases: always needed) and one tag's values in ONE external block - RN cannot be skipped, so its block is needed, so the tag that
    shares it stays as well; every other tag goes. Before the fix RN's blocks never reached the needed set, the tag was dropped, its block was not inflated and RN was
    decoded from an empty block: a CramError on a valid file."""
    twin = twins["MappingQC_in2.bam"]; src = str(tmp_path / "shared.cram"); out = str(tmp_path / "s.bam")
    monkeypatch.setattr(CE, "SHARED_BLOCK", True)
    CE.encode(twin["bam"], src, twin["genome"])
    fetch = cram_twin.ref_fetch_of(twin); ngsqc.set_reference(twin["fasta"]); monkeypatch.delenv("NGSQC_CRAM_NO_REFERENCE", raising=False)
    try:
        f = CD.read_cram(src, fetch); rgs = CD.read_groups(f.header)
        ngsqc.set_cram_skip(ngsqc.CRAM_SKIP_NAMES | ngsqc.CRAM_SKIP_TAGS)
        try:
            ngsqc.cram_to_bam(src, out)
        finally:
            ngsqc.set_cram_skip(0)
        _, _, recs = split_bam(bam_stream(out)[0])
        assert len(recs) == len(f.records) > 1000
        n_tagged = 0
        for got, r in zip(recs, f.records):
            full = CD.to_bam_record(r, rgs)
            assert got[4:36 + len(r.name) + 1] == full[4:36 + len(r.name) + 1]   # the names stayed (their block is needed); [0:4] is block_size: most tags are gone
            kept = [t for t in r.tags if bytes(t[0]) in got[36:]]; n_tagged += bool(kept)
        assert n_tagged > 0
    finally:
        ngsqc.set_reference(None)


def test_a_threads_own_skip_choice_is_its_own(tmp_path):
    """ngsqc_set_cram_skip_thread: a scope's choice holds for the calling thread only and gives back what was set before (the guard of Statistics::mapping)."""
    import threading
    ngsqc.set_cram_skip(ngsqc.CRAM_SKIP_NAMES | ngsqc.CRAM_SKIP_TAGS)
    try:
        assert ngsqc.set_cram_skip_thread(ngsqc.CRAM_SKIP_NAMES) == -1
        seen = []
        t = threading.Thread(target=lambda: seen.append(ngsqc.set_cram_skip_thread(-1))); t.start(); t.join()
        assert seen == [-1]                                                       # the other thread never had a choice of its own
        assert ngsqc.set_cram_skip_thread(0) == ngsqc.CRAM_SKIP_NAMES             # ours is still there
        assert ngsqc.set_cram_skip_thread(-1) == 0
        assert ngsqc.set_cram_skip_thread(64) == -3                               # NGSQC_E_ARG: not a flag
        assert ngsqc.set_cram_skip_thread(-1) == -1                               # ... and nothing was set
    finally:
        ngsqc.set_cram_skip_thread(-1); ngsqc.set_cram_skip(0)


def test_genome_errors(twins, tmp_path, monkeypatch):
    twin = twins["MappingQC_in2.bam"]; cram = str(tmp_path / "twin.cram"); out = str(tmp_path / "o.bam")
    CE.encode(twin["bam"], cram, twin["genome"])
    monkeypatch.delenv("NGSQC_CRAM_NO_REFERENCE", raising=False); monkeypatch.delenv("NGSQC_REFERENCE", raising=False)
    # the genome through the environment
    monkeypatch.setenv("NGSQC_REFERENCE", twin["fasta"]); ngsqc.set_reference(None)
    ngsqc.cram_to_bam(cram, out)
    monkeypatch.delenv("NGSQC_REFERENCE")
    # no genome, a path that does not exist: BamReader.cpp:486-489
    for ref in (None, str(tmp_path / "nothing.fa")):
        ngsqc.set_reference(ref)
        with pytest.raises(ngsqc.NgsqcError) as e:
            ngsqc.cram_to_bam(cram, out)
        assert "Error while setting reference genome" in str(e.value)
    # another genome of the same lengths: the slices' MD5 does not match (htslib: "md5sum reference mismatch")
    used = next(n for n, l in twin["refs"] if l > 1000)
    other = str(tmp_path / "other.fa")
    with open(other, "wb") as f, open(other + ".fai", "w") as fai:
        for n, l in twin["refs"]:
            f.write(b">" + n.encode() + b"\n"); off = f.tell()
            g = bytearray(twin["genome"][n])
            if n == used: g[150] = ord("A") if g[150] != ord("A") else ord("C")
            f.write(bytes(g) + b"\n"); fai.write("%s\t%d\t%d\t%d\t%d\n" % (n, l, off, l, l + 1))
    ngsqc.set_reference(other)
    with pytest.raises(ngsqc.NgsqcError) as e:
        ngsqc.cram_to_bam(cram, out)
    assert "md5sum reference mismatch" in str(e.value)
    monkeypatch.setenv("NGSQC_CRAM_IGNORE_MD5", "1")
    ngsqc.cram_to_bam(cram, out)                        # (htslib's ignore_md5 option)
    monkeypatch.delenv("NGSQC_CRAM_IGNORE_MD5")
    # a genome whose contig has another length: BamReader::checkChromosomeLengths (BamReader.cpp:491)
    short = str(tmp_path / "short.fa")
    with open(short, "wb") as f, open(short + ".fai", "w") as fai:
        for n, l in twin["refs"]:
            l2 = l - 5 if n == used else l
            f.write(b">" + n.encode() + b"\n"); off = f.tell(); f.write(twin["genome"][n][:l2] + b"\n"); fai.write("%s\t%d\t%d\t%d\t%d\n" % (n, l2, off, l2, l2 + 1))
    ngsqc.set_reference(short)
    with pytest.raises(ngsqc.NgsqcError) as e:
        ngsqc.cram_to_bam(cram, out)
    assert "differs from the length" in str(e.value)
    ngsqc.set_reference(None)


def test_region_selection_decodes_only_the_overlapping_slices(twins, tmp_path):
    """what ngsqc_open_regions does with a CRAM: slices whose headers overlap a region (read from the slice headers, no .crai needed) - every record that overlaps the
    region is there, in file order, and far fewer records are decoded than the file holds"""
    twin = twins["MappingQC_in5.bam"]; cram = str(tmp_path / "twin.cram"); out = str(tmp_path / "part.bam")
    CE.encode(twin["bam"], cram, twin["genome"], slice_records=200)
    ngsqc.set_reference(twin["fasta"])
    try:
        parsed = [CE.parse_record(r) for r in twin["records"]]
        names = [n for n, _ in twin["refs"]]
        used = sorted({r["ref_id"] for r in parsed if r["ref_id"] >= 0 and not r["flag"] & 4})
        t = used[len(used) // 2]; on_t = [r for r in parsed if r["ref_id"] == t]
        mid = on_t[len(on_t) // 2]["pos"]; region = (names[t], mid, mid + 150)
        bare = region[0][3:] if region[0].startswith("chr") else region[0]
        for chrom in (region[0], bare if region[0].startswith("chr") else "chr" + bare, "CHR" + bare.upper(), bare.lower()):   # the reference's Chromosome rule: "chr" / "CHR" dropped, upper case (as the BAM index path)
            ngsqc.cram_to_bam(cram, out, regions=[(chrom, region[1], region[2])])
            _, _, recs = split_bam(bam_stream(out)[0])
            want = [r["raw"] for r in parsed if r["ref_id"] == t and r["pos"] <= region[2] and CE.ref_end(r) >= region[1]]
            assert want and 0 < len(recs) < len(parsed) // 3
            it = iter(recs)
            assert all(any(w == g for g in it) for w in want)       # all of them, in order
        ngsqc.cram_to_bam(cram, out, regions=[("no_such_contig", 1, 100)])
        assert split_bam(bam_stream(out)[0])[2] == []
    finally:
        ngsqc.set_reference(None)


# ---- the quality arrays on the device (csrc/cram_dev.hip): the host leaves the QS blocks compressed and writes a plan; here the plan is replayed in Python the way the
# kernels do it (compact cumulative tables, linear search, four states over one byte stream; then the per-record copy) - the GPU tests run the kernels themselves ----
import numpy as np  # noqa: E402


def _replay_plan(cram_bytes, plan_path, stream):
    d = open(plan_path, "rb").read()
    nj, nt, ns_, npch, out_bytes = struct.unpack_from("<5Q", d, 0); o = 40
    jobs = [struct.unpack_from("<QQIIIIII", d, o + 40 * i) for i in range(nj)]; o += 40 * nj
    tabs = np.frombuffer(d, dtype="<u2", count=nt, offset=o); o += 2 * nt
    syms = d[o:o + ns_]; o += ns_
    patches = [struct.unpack_from("<QQII", d, o + 24 * i) for i in range(npch)]
    qs = bytearray(out_bytes)
    for in_off, out_off, in_len, n_out, tab_off, sym_off, order, ns in jobs:
        p = in_off; end = in_off + in_len; sym = syms[sym_off:sym_off + 64]; lut = syms[sym_off + 64:sym_off + 320]; row = ns + 1
        R = list(struct.unpack_from("<4I", cram_bytes, p)); p += 16

        def step(j, C0):
            nonlocal p
            m = R[j] & 0xfff; k = 0
            while k + 1 < ns and tabs[C0 + k + 1] <= m: k += 1
            c0 = int(tabs[C0 + k]); f = int(tabs[C0 + k + 1]) - c0
            assert f > 0 and c0 <= m < c0 + f
            v = f * (R[j] >> 12) + m - c0
            while v < (1 << 23): assert p < end; v = (v << 8) | cram_bytes[p]; p += 1
            R[j] = v
            return k
        if order == 0:
            for i in range(n_out): qs[out_off + i] = sym[step(i & 3, tab_off)]
        else:
            q = n_out >> 2; idx = [0, q, 2 * q, 3 * q]; pk = [lut[0]] * 4; assert lut[0] < ns
            for _ in range(q):
                for j in range(4):
                    k = step(j, tab_off + pk[j] * row); qs[out_off + idx[j]] = sym[k]; idx[j] += 1; pk[j] = k
            while idx[3] < n_out:
                k = step(3, tab_off + pk[3] * row); qs[out_off + idx[3]] = sym[k]; idx[3] += 1; pk[3] = k
    out = bytearray(stream)
    for dst, src, ln, _ in patches:
        assert all(b == 0 for b in out[dst:dst + ln]) and src + ln <= out_bytes
        out[dst:dst + ln] = qs[src:src + ln]
    return bytes(out), nj, npch


@pytest.mark.parametrize("case", ["SampleIdentity_in_rna.cram", "cramTest.cram", "twin", "twin_order0", "twin_plain"])
def test_device_quality_plan_replayed(case, twins, tmp_path, monkeypatch):
    if case.startswith("twin"):
        twin = twins["MappingQC_in5.bam"]; src = str(tmp_path / "t.cram")
        kw = {"twin": {}, "twin_order0": dict(methods=[4]), "twin_plain": dict(variety=False)}[case]
        CE.encode(twin["bam"], src, twin["genome"], slice_records=1200, **kw); ngsqc.set_reference(twin["fasta"])
    else:
        src = os.path.join(GI, case); monkeypatch.setenv("NGSQC_CRAM_NO_REFERENCE", "1")
    try:
        full = str(tmp_path / "full.bam"); blank = str(tmp_path / "blank.bam"); plan = str(tmp_path / "plan.bin")
        ngsqc.cram_to_bam(src, full)
        monkeypatch.setenv("NGSQC_CRAM_PLAN_DUMP", plan)
        ngsqc.cram_to_bam(src, blank)
        monkeypatch.delenv("NGSQC_CRAM_PLAN_DUMP")
    finally:
        ngsqc.set_reference(None)
    want = bam_stream(full)[0]; got, n_jobs, n_patches = _replay_plan(open(src, "rb").read(), plan, bam_stream(blank)[0])
    assert got == want
    if case == "twin_plain": assert n_jobs == 0 and n_patches == 0        # raw quality blocks: nothing to decode
    else: assert n_jobs >= 1 and n_patches > 500 and bam_stream(blank)[0] != want


def test_empty_cram_and_missing_eof_container(twins, tmp_path):
    twin = twins["MappingQC_in2.bam"]
    # a file without records: header container + EOF container
    empty_bam = str(tmp_path / "empty.bam"); cram_twin.write_bam(empty_bam, twin["text"], twin["refs"], [])
    cram = str(tmp_path / "empty.cram"); out = str(tmp_path / "o.bam")
    CE.encode(empty_bam, cram, twin["genome"])
    ngsqc.cram_to_bam(cram, out)
    text, refs, recs = split_bam(bam_stream(out)[0])
    assert text.decode() == twin["text"] and refs == twin["refs"] and recs == []
    # the EOF container cut off (a file that was still being written): htslib warns and reads what is there
    full = str(tmp_path / "full.cram"); CE.encode(twin["bam"], full, twin["genome"], rr=False)
    d = open(full, "rb").read(); assert d[-38:] == CE.eof_container()
    cut = str(tmp_path / "noeof.cram"); open(cut, "wb").write(d[:-38])
    ngsqc.cram_to_bam(cut, out)
    assert split_bam(bam_stream(out)[0])[2] == twin["records"]


def test_lossy_quality_records_fall_back_to_the_host(twins, tmp_path, monkeypatch):
    """records without a quality array whose qualities come as features (Q: one, q: a stretch; the rest stays 0xff - lossy-quality CRAMs): the product equals the oracle;
    such a record reads SINGLE bytes out of the QS block, so a slice that holds one keeps its block on the host (no job in the device plan) and still gives the full records"""
    twin = twins["MappingQC_in2.bam"]; cram = str(tmp_path / "lossy.cram"); out = str(tmp_path / "o.bam"); blank = str(tmp_path / "b.bam"); plan = str(tmp_path / "p.bin")
    CE.encode(twin["bam"], cram, twin["genome"], qual_features=True, slice_records=700)
    f = CD.read_cram(cram, cram_twin.ref_fetch_of(twin)); rgs = CD.read_groups(f.header)
    want = [CD.to_bam_record(r, rgs) for r in f.records]
    lossy = [r for r in f.records if not r.cf & CD.CF_QUAL_ARRAY and any(c in "Qq" for c, _, _ in r.features)]
    assert len(lossy) > 100 and all(r.qual[2] != 0xff and r.qual[4] == 0xff and r.qual[9:17] != b"\\xff" * 8 for r in lossy)
    ngsqc.set_reference(twin["fasta"])
    try:
        ngsqc.cram_to_bam(cram, out)
        monkeypatch.setenv("NGSQC_CRAM_PLAN_DUMP", plan); ngsqc.cram_to_bam(cram, blank); monkeypatch.delenv("NGSQC_CRAM_PLAN_DUMP")
    finally:
        ngsqc.set_reference(None)
    assert split_bam(bam_stream(out)[0])[2] == want
    got, n_jobs, n_patches = _replay_plan(open(cram, "rb").read(), plan, bam_stream(blank)[0])
    assert split_bam(got)[2] == want
    n_slices = sum(len(ss) for _, ss, _ in f.containers)
    assert n_jobs < n_slices          # slices with such records are not in the plan
