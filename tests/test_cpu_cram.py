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
    v31 = bytearray(d); v31[5] = 1
    p = str(tmp_path / "v31.cram"); open(p, "wb").write(v31)
    with pytest.raises(ngsqc.NgsqcError) as e:
        ngsqc.cram_to_bam(p, out)
    assert e.value.code == ngsqc.capi.E_UNSUPPORTED if hasattr(ngsqc, "capi") and hasattr(ngsqc.capi, "E_UNSUPPORTED") else "3.1" in str(e.value)
    with pytest.raises(ngsqc.NgsqcError):
        ngsqc.cram_to_bam(os.path.join(GI, "sry.bam"), out)
