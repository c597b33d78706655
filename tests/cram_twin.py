"""Test helper: a BAM / CRAM / genome triple ("twin") built from one of the reference's BAM fixtures. The fixture's reads are moved onto short contigs (a window of
each chromosome, positions shifted: the real contigs are hundreds of Mb and no genome for them is in the tree), a genome is made up for those contigs from the
reads themselves plus noise (so that most bases match it and some do not), and oracle/cram_encode.py writes the CRAM. Reading the CRAM must give back the BAM."""
import os
import random
import struct
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import cram_decode as CD  # noqa: E402
import cram_encode as CE  # noqa: E402


def _bgzf(raw, level=6):
    c = zlib.compressobj(level, zlib.DEFLATED, -15); body = c.compress(raw) + c.flush()
    return b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body + struct.pack("<II", zlib.crc32(raw), len(raw))


def write_bam(path, text, refs, raw_records):
    hdr = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs))
    for name, ln in refs: hdr += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
    out = bytearray(); chunk = bytearray(hdr)          # (a header of thousands of contigs spans several members, like the records)
    for r in raw_records:
        chunk += r
        while len(chunk) > 60000: out += _bgzf(bytes(chunk[:60000])); del chunk[:60000]
    if chunk: out += _bgzf(bytes(chunk))
    open(path, "wb").write(bytes(out) + _bgzf(b""))


def make_twin(src_bam, out_dir, window=250_000, seed=1, max_records=None):
    """-> dict(bam, fasta, genome, text, refs, records (raw bytes))"""
    text, refs, recs = CE.read_bam(src_bam)
    first = {}
    for r in recs:
        if r["ref_id"] >= 0 and r["pos"] >= 1: first.setdefault(r["ref_id"], r["pos"])
    keep = []
    for r in recs:
        t = r["ref_id"]
        if t < 0: keep.append(r); continue
        if r["pos"] < 1 or r["pos"] >= first[t] + window: continue
        keep.append(r)
    if max_records: keep = keep[:max_records]
    shift = {t: p - 101 for t, p in first.items()}
    new_len = {}
    for r in keep:
        t = r["ref_id"]
        if t >= 0: new_len[t] = max(new_len.get(t, 0), CE.ref_end(r) - shift[t] + 300)
    new_refs = [(n, new_len.get(i, 1000)) for i, (n, _) in enumerate(refs)]
    lines = []
    for ln in text.split("\n"):
        if ln.startswith("@SQ"):
            f = ln.split("\t"); name = [x for x in f if x.startswith("SN:")][0][3:]
            i = [n for n, _ in refs].index(name)
            f = [("LN:%d" % new_refs[i][1]) if x.startswith("LN:") else x for x in f if not x.startswith(("M5:", "UR:"))]
            ln = "\t".join(f)
        lines.append(ln)
    new_text = "\n".join(lines)
    rng = random.Random(seed)
    genome = {n: bytearray(rng.choice(b"ACGT") for _ in range(l)) if i in new_len else bytearray(b"ACGT" * (l // 4 + 1))[:l] for i, (n, l) in enumerate(new_refs)}
    seen = {n: bytearray(l) for n, l in new_refs}
    raws = []
    for r in keep:
        t = r["ref_id"]; pos = r["pos"] - shift[t] if t >= 0 else r["pos"]
        mt = r["mate_ref"]; mpos = r["mate_pos"]
        if mt >= 0 and mt in shift: mpos = max(0, mpos - shift[mt])
        end = pos
        if t >= 0 and not r["flag"] & 4:
            g = genome[refs[t][0]]; s = seen[refs[t][0]]; rp = 0; gp = pos - 1
            for op, k in r["cigar"]:
                if op in "M=X":
                    for x in range(k):
                        if gp + x < len(g) and not s[gp + x] and r["seq"] and chr(r["seq"][rp + x]) in "ACGT": g[gp + x] = r["seq"][rp + x]; s[gp + x] = 1
                    rp += k; gp += k
                elif op in "IS": rp += k
                elif op in "DN": gp += k
            end = gp if r["cigar"] else pos
        raw = bytearray(r["raw"])
        pos0 = pos - 1; end0 = end if (t >= 0 and not r["flag"] & 4 and r["cigar"]) else pos0 + 1
        struct.pack_into("<i", raw, 8, pos0); struct.pack_into("<H", raw, 14, CD.reg2bin(pos0, end0) if pos0 >= 0 else 4680); struct.pack_into("<i", raw, 28, mpos - 1)
        raws.append(bytes(raw))
    # a few genome positions that no read agrees with, some N, and lower case in the FASTA (readers fold case)
    for n, g in genome.items():
        for _ in range(len(g) // 700): g[rng.randrange(len(g))] = rng.choice(b"ACGTN")
    os.makedirs(out_dir, exist_ok=True)
    bam = os.path.join(out_dir, "twin.bam"); fasta = os.path.join(out_dir, "genome.fa")
    write_bam(bam, new_text, new_refs, raws)
    with open(fasta, "wb") as f, open(fasta + ".fai", "w") as fai:
        for n, l in new_refs:
            f.write(b">" + n.encode() + b" made up\n"); off = f.tell(); g = bytes(genome[n])
            body = bytearray()
            for o in range(0, l, 60):
                line = g[o:o + 60]
                body += (line.lower() if (o // 60) % 7 == 3 else line) + b"\n"
            f.write(body); fai.write("%s\t%d\t%d\t60\t61\n" % (n, l, off))
    return dict(bam=bam, fasta=fasta, genome={n: bytes(g) for n, g in genome.items()}, text=new_text, refs=new_refs, records=raws)


def ref_fetch_of(twin):
    names = [n for n, _ in twin["refs"]]
    return lambda ref_id, p0, n: twin["genome"][names[ref_id]][p0:p0 + n]


def long_cigar_bam(path, n_groups=17500, seed=5):
    """A small coordinate-sorted BAM on one 200 kb contig with ordinary reads around ONE read of 4 * n_groups CIGAR operations (2M 1I 2M 1D, repeated): more
    than 65535 do not fit n_cigar_op, so the record carries the BAM convention - the placeholder <l_seq>S<reference length>N and a CG:B,I tag behind its other
    tags (SAM spec 4.2.2). -> the raw records"""
    rng = random.Random(seed)
    text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chrL\tLN:200000\n@PG\tID:x\tPN:x\n"
    refs = [("chrL", 200000)]

    def rec(name, pos0, ops, flag=0, mapq=60, tags=b"NMC\x03"):
        l_seq = sum(k for op, k in ops if op in "MIS=X"); ref_len = sum(k for op, k in ops if op in "MDN=X")
        seq = bytes(rng.choice(b"ACGT") for _ in range(l_seq)); qual = bytes(rng.choice((2, 12, 23, 37)) for _ in range(l_seq))
        code = {c: i for i, c in enumerate(b"=ACMGRSVTWYHKDBN")}; packed = bytearray((l_seq + 1) // 2)
        for i, ch in enumerate(seq): packed[i >> 1] |= code[ch] << 4 if not i & 1 else code[ch]
        cig = b"".join(struct.pack("<I", k << 4 | "MIDNSHP=X".index(op)) for op, k in ops); n_cig = len(ops)
        if n_cig > 65535:
            tags = tags + b"CGBI" + struct.pack("<i", n_cig) + cig
            cig = struct.pack("<II", l_seq << 4 | 4, ref_len << 4 | 3); n_cig = 2
        nm = name + b"\0"
        body = struct.pack("<iiBBHHHiiii", 0, pos0, len(nm), mapq, CD.reg2bin(pos0, pos0 + max(ref_len, 1)), n_cig, flag, l_seq, -1, -1, 0) + nm + cig + bytes(packed) + qual + tags
        return struct.pack("<i", len(body)) + body
    raws = [rec(b"short%04d" % i, 100 + 37 * i, [("M", 150)]) for i in range(40)]
    raws.append(rec(b"long_cigar", 1700, [("S", 7)] + [("M", 2), ("I", 1), ("M", 2), ("D", 1)] * n_groups + [("S", 5)], flag=16))
    raws += [rec(b"after%04d" % i, 1800 + 911 * i, [("M", 100), ("D", 2), ("M", 50)]) for i in range(100)]
    write_bam(path, text, refs, raws)
    return raws
