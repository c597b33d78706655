"""Pins oracle/cram_decode.py (the CPU restatement of CRAM 3.0 decoding, the checker of the product's CRAM input) on the reference's own CRAM fixtures and on the
known answers of the reference's CRAM tests (src/cppNGS-TEST/BamReader_Test.cpp:400-560) that do not depend on the hg38 genome. No GPU."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GI = os.path.join(HERE, "golden", "ref_in")
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import cram_decode as CD  # noqa: E402


@pytest.fixture(scope="module")
def cram_test():
    return CD.read_cram(os.path.join(GI, "cramTest.cram"))


def test_structure_and_checksums(cram_test):
    """every block and container header CRC-32 is checked while reading; the containers' record counts equal their slices' and the decoded records"""
    f = cram_test
    assert f.version == (3, 0) and f.eof
    assert len(f.records) == 36197 == sum(k.n_records for k, _, _ in f.containers) == sum(s.n_records for _, ss, _ in f.containers for s in ss)
    names = CD.ref_names(f.header)
    assert names[0] == ("chr1", 248956422) and len(names) > 20
    for k, slices, ch in f.containers:
        assert ch.RN
        for s in slices:
            assert s.ref_id >= -2 and s.n_blocks == len(s.content_ids) + 1


def test_first_properly_paired_read(cram_test):
    """BamReader_Test.cpp CramSupport_referenceAsParameter_tests / CramSupport_skippedFields: everything but the bases (they come from hg38, which is not in the tree)"""
    names = CD.ref_names(cram_test.header)
    r = next(r for r in cram_test.records if r.bf & CD.BAM_FPROPER)
    assert r is next(r for r in cram_test.records if not r.bf & CD.BAM_FUNMAP)
    assert r.name == b"PC0226:121:000000000-AB2J9:1:2101:19474:26718"
    assert names[r.ref_id][0] == "chr1" and r.pos == 1008789 and r.end == 1008918 and r.rl == 130
    assert CD.cigar_string(r) == "130M"
    assert r.tlen == 130 and r.mapq == 60
    assert names[r.mate_ref][0] == "chr1" and r.mate_pos == 1008789
    assert bytes(q + 33 for q in r.qual) == b"3>AABF@FFFFFGGGGGGGGGFHHHFGGGCGGGGEEGGGGHCGHHHHHHHHGHHHGHGFGHHHHGGGGGGHHHHHHHHGFGGGGGHHFEHFHGHHHHHHHGHGGGHHGGFGGGHHHFHHHHHHHHGGFGG"
    tags = {t: (typ, v) for t, typ, v in r.tags}
    assert tags[b"MC"] == (ord("Z"), b"130M\0") and tags[b"AS"][1] == bytes([130])
    assert b"RG" not in tags and r.rg == -1          # al.tag("RG") == ""


def test_cigars_of_the_second_mapped_and_the_last_read(cram_test):
    """CramSupport_cigarDataAsString"""
    mapped = [r for r in cram_test.records if not r.bf & CD.BAM_FUNMAP]
    assert CD.cigar_string(mapped[1]) == "130M"
    assert CD.cigar_string(cram_test.records[-1]) == "19S139M"


def _base_at(r, pos):
    """BamAlignment::extractBaseByCIGAR (BamReader.cpp:307-370): (kind, quality, index in the read); kind 'b' = a base of the read, '-' deleted, '~' skipped / clipped"""
    if all(op == "I" for op, _ in r.cigar): return "~", -1, -1
    read_pos = 0; genome_pos = r.pos - 1
    for op, n in r.cigar:
        if op in "M=X": genome_pos += n; read_pos += n
        elif op == "I": read_pos += n
        elif op == "D":
            genome_pos += n
            if genome_pos >= pos: return "-", 255, -1
        elif op == "N":
            genome_pos += n
            if genome_pos >= pos: return "~", -1, -1
        elif op == "S":
            read_pos += n
            if read_pos >= r.rl: return "~", -1, -1
        if genome_pos >= pos:
            i = read_pos - (genome_pos - pos) - 1
            return "b", r.qual[i], i
    return "~", -1, -1


def _pileup(f, chrom, pos):
    """BamReader::getPileup with its defaults (min_mapq 1, properly paired only, min_baseq 13): depth of A+C+G+T, and the reads whose base at pos is a substitution feature"""
    tid = [n for n, _ in CD.ref_names(f.header)].index(chrom); depth = 0; subst = 0
    for r in f.records:
        if r.ref_id != tid or r.bf & (0x100 | 0x800 | 0x400 | 4) or not r.bf & 2 or r.mapq < 1: continue
        if not (r.pos <= pos <= r.end): continue
        kind, q, i = _base_at(r, pos)
        if kind == "b" and q >= 13:
            depth += 1
            if any(c == "X" and fp - 1 == i for c, fp, _ in r.features): subst += 1
    return depth, subst


def test_pileup_depths(cram_test):
    """CramSupport_getPileup: the depths need positions, CIGARs, flags, mapping and base qualities of hundreds of reads; the allele split of a SNP is the number of reads
    that carry a substitution feature there (frequency("G", "A") = 0.508876 = 86 of 169)"""
    assert _pileup(cram_test, "chr1", 27355990) == (169, 86)
    d, s = _pileup(cram_test, "chr1", 27359572); assert d == 175 and s in (0, 175)      # homozygous: every read or none differs from the reference
    assert _pileup(cram_test, "chr1", 27360975) == (736, 380)                            # 0.516304 * 736
    assert _pileup(cram_test, "chr1", 27363868)[0] == 111
    assert _pileup(cram_test, "chr3", 10052522)[0] == 25
    assert _pileup(cram_test, "chr2", 47806751)[0] == 32
    assert _pileup(cram_test, "chr6", 130827751)[0] == 703
    assert _pileup(cram_test, "chr5", 80864777)[0] == 16


def _indels(f, chrom, pos, window):
    """the indel part of BamReader::getPileup (BamReader.cpp:870-877) with BamAlignment::extractIndelsByCIGAR (:376-438): (insertions, deletions) of the reads that
    overlap pos and pass the pileup's filters, within +- window of pos"""
    tid = [n for n, _ in CD.ref_names(f.header)].index(chrom); ins = dele = 0
    for r in f.records:
        if r.ref_id != tid or r.bf & (0x100 | 0x800 | 0x400 | 4) or not r.bf & 2 or r.mapq < 1: continue
        if not (r.pos <= pos <= r.end): continue
        g = r.pos
        for op, n in r.cigar:
            if op in "M=X": g += n
            elif op == "I":
                if pos - window <= g <= pos + window: ins += 1
            elif op == "D":
                if pos - window <= g <= pos + window: dele += 1
                g += n
            elif op == "N": g += n
            if g > pos + window: break
    return ins, dele


def test_pileup_indels(cram_test):
    """CramSupport_getPileup, the indel counts: positions of I and D inside the CIGARs that the decoder builds from the read features of hundreds of reads"""
    assert _indels(cram_test, "chr3", 10052522, 1) == (10, 4)           # indels().count() == 14: 10 with '+', 4 with '-'
    assert _indels(cram_test, "chr2", 47806751, 1) == (12, 14)          # 26, 14 with '-'
    assert _indels(cram_test, "chr6", 130827751, 3) == (298, 27)        # 325
    assert _indels(cram_test, "chr5", 80864777, 4) == (0, 6)            # 6, all deletions
    for chrom, pos in (("chr1", 27355990), ("chr1", 27359572), ("chr1", 27360975), ("chr1", 27363868)):
        assert _indels(cram_test, chrom, pos, 1) == (0, 0)


@pytest.mark.parametrize("name,n_records,rr", [("SampleIdentity_in_wes.cram", 17534, True), ("SampleIdentity_in_rna.cram", 10528, False)])
def test_other_fixtures_decode(name, n_records, rr):
    f = CD.read_cram(os.path.join(GI, name))
    assert len(f.records) == n_records and f.eof and all(ch.RR == rr for _, _, ch in f.containers)
    mapped = [r for r in f.records if not r.bf & 4]
    assert all(sum(n for op, n in r.cigar if op in "MIS=X") == r.rl for r in mapped)       # the CIGAR spans the read
    assert all(r.end - r.pos + 1 == sum(n for op, n in r.cigar if op in "MDN=X") for r in mapped)
    if not rr:   # every base is stored in the file: nothing comes from a genome
        assert not any(r.bases_from_ref for r in f.records)
        assert all(set(r.seq) <= set(b"ACGTN") for r in mapped)
    # positions ascend inside a reference (a coordinate-sorted file)
    last = {}
    for r in mapped:
        assert last.get(r.ref_id, 0) <= r.pos; last[r.ref_id] = r.pos


def test_mate_chains_agree_with_the_mc_tag():
    """the CIGAR of the mate that bwa wrote into MC equals the CIGAR decoded for the record the chain (NF) points at, and the template lengths of a pair are opposite"""
    f = CD.read_cram(os.path.join(GI, "cramTest.cram"))
    # chains are resolved per slice: walk the slices
    start = 0; checked = 0
    for k, slices, ch in f.containers:
        for s in slices:
            recs = f.records[start:start + s.n_records]; start += s.n_records
            for i, r in enumerate(recs):
                if r.cf & CD.CF_DETACHED or r.nf is None: continue
                m = recs[i + r.nf + 1]
                assert m.name == r.name and m.mate_pos == r.pos and r.mate_pos == m.pos and r.mate_ref == m.ref_id
                tags = {t: v for t, _, v in r.tags}
                if b"MC" in tags and not m.bf & 4:
                    assert tags[b"MC"] == CD.cigar_string(m).encode() + b"\0"; checked += 1
                if r.ref_id == m.ref_id and not (r.bf | m.bf) & 4 and recs[i].mate_line == i + r.nf + 1 and m.nf is None:
                    assert r.tlen == -m.tlen
                assert bool(r.bf & 0x20) == bool(m.bf & 0x10) and bool(m.bf & 0x20) == bool(r.bf & 0x10)
    assert checked > 10000


# ---- the CRAM writer of the oracle (oracle/cram_encode.py: what gives the product's CRAM input a BAM truth): read back by the pinned decoder it returns the BAM's records ----
import cram_encode as CE  # noqa: E402
import cram_twin  # noqa: E402

VARIANTS = {"default": {}, "no_genome_needed": dict(rr=False), "multi_reference_slices": dict(multi_ref=True, slice_records=700), "embedded_reference": dict(embed_ref=True),
            "plain_external": dict(variety=False, chains=False), "small_slices": dict(slice_records=150), "bzip2_and_lzma_blocks": dict(methods=[2, 3, 1]),
            "containers_of_three_slices": dict(slices_per_container=3, slice_records=300), "containers_of_mixed_slices": dict(slices_per_container=4, slice_records=250, multi_ref=True),
            # CRAM 3.1: every shape of the rANS Nx16 codec over the external blocks, and a mix with the 3.0 methods
            "cram31_rans_nx16": dict(version=(3, 1), methods=[50, 51, 52, 53, 54, 55, 56, 57, 58, 59]), "cram31_mixed_methods": dict(version=(3, 1), methods=[51, 1, 58, 41, 56, 53], slice_records=400)}


def test_eof_container_equals_the_fixtures():
    tail = open(os.path.join(GI, "cramTest.cram"), "rb").read()[-38:]
    assert CE.eof_container() == tail == open(os.path.join(GI, "SampleIdentity_in_rna.cram"), "rb").read()[-38:]


def test_rans_nx16_round_trips():
    """the CRAM 3.1 codec of the oracle against its own writer (no htslib-written 3.1 file exists in the reference: unpinned, see oracle/cram_decode.py): every transform
    alone and stacked, sizes around the interleave widths, and the coded size of random ACGT (2 bits per base: the entropy coder is one)"""
    import random
    rng = random.Random(4)
    shapes = [{}, dict(order=1), dict(x32=True), dict(order=1, x32=True), dict(order=1, comp_table=True, shift=10), dict(pack=True), dict(rle=True), dict(order=1, rle=True, comp_meta=True),
              dict(rle=True, pack=True), dict(stripe=4), dict(stripe=3, order=1), dict(cat=True), dict(half_total=True), dict(order=1, x32=True, comp_table=True, rle=True, pack=True)]
    q = bytearray(); v = 30
    for _ in range(9000):
        if rng.random() < 0.05: v = rng.choice([2, 12, 23, 37, 40])
        q.append(v)
    datas = [b"", b"A", b"A" * 44, bytes(q), bytes([7]) * 3 + bytes([9]) * 30000, b"".join(int(rng.gauss(1000, 300)).to_bytes(4, "little", signed=True) for _ in range(500))]
    for n in (3, 4, 5, 31, 32, 33, 127, 128, 129, 2000):
        datas += [bytes(rng.randrange(256) for _ in range(n)), bytes(rng.choice(b"ACGT") for _ in range(n)), bytes(rng.choice(b"AC") for _ in range(n))]
    for d in datas:
        for kw in shapes:
            c = CE.rans_nx16_encode(d, **kw); cur = CD.Cursor(c)
            assert CD.rans_nx16_decode(cur) == d and cur.p == len(c), (len(d), kw)
    acgt = bytes(rng.choice(b"ACGT") for _ in range(40000))
    assert 10000 <= len(CE.rans_nx16_encode(acgt)) <= 10000 + 200 and len(CE.rans_nx16_encode(bytes([30]) * 5000 + bytes([2]) * 3000, rle=True, pack=True)) < 64


def test_rans_encoder_round_trips():
    import random
    rng = random.Random(2)
    for n in (0, 1, 3, 4, 5, 17, 1000, 70001):
        for data in (bytes(rng.randrange(256) for _ in range(n)), bytes(rng.choice(b"AAAAACGT") for _ in range(n)), bytes([7]) * n):
            for order in (0, 1):
                assert CD.rans_decode(CE.rans_encode(data, order)) == data


@pytest.mark.parametrize("src", ["MappingQC_in2.bam", "BamReader_rna.bam"])
@pytest.mark.parametrize("variant", list(VARIANTS))
def test_written_cram_reads_back_as_the_bam(src, variant, tmp_path):
    twin = cram_twin.make_twin(os.path.join(GI, src), str(tmp_path), max_records=4000)
    cram = str(tmp_path / "twin.cram")
    CE.encode(twin["bam"], cram, twin["genome"], **VARIANTS[variant])
    f = CD.read_cram(cram, cram_twin.ref_fetch_of(twin))
    rgs = CD.read_groups(f.header)
    assert f.header == twin["text"] and len(f.records) == len(twin["records"]) > 500
    assert not any(r.bases_from_ref for r in f.records)
    for r, raw in zip(f.records, twin["records"]):
        assert CD.to_bam_record(r, rgs) == raw, (r.name, CD.to_bam_record(r, rgs)[:48].hex(), raw[:48].hex())
    kinds = {c for r in f.records for c, _, _ in r.features}
    if variant == "default":
        assert {"X", "S"} <= kinds and any(r.nf is not None for r in f.records) and any(r.cf & CD.CF_DETACHED for r in f.records)
    if variant == "no_genome_needed": assert "b" in kinds and "X" not in kinds
