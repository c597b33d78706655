"""ORACLE (test infrastructure, never imported by the product): CPU restatement of CRAM 3.0 decoding - file structure, codecs, record decode and the
reconstruction of BAM records (hts-specs CRAMv3; the reference reads CRAM through htslib's sam_read1 under BamReader, src/cppNGS/BamReader.cpp:482-492,
525-572, with the reference genome the user names). htslib is a third-party dependency that is absent from /root/reference, so the published format is
restated here:
  * file definition, containers, blocks (CRC-32 of every container header and block is CHECKED), ITF8 / LTF8
  * block methods raw (0), gzip (1), bzip2 (2), lzma (3), rANS 4x8 order 0 / order 1 (4); of CRAM 3.1: rANS Nx16 (5) with all its transforms - UNPINNED, see there
  * compression header: preservation map (RN, AP, RR, SM, TD), data-series encodings, tag encodings
  * encodings NULL, EXTERNAL, HUFFMAN, BYTE_ARRAY_LEN, BYTE_ARRAY_STOP, BETA, SUBEXP, GAMMA
  * slice header, records, read features -> CIGAR / bases / qualities, mate chains inside a slice (NF) and detached mates, template length rules
PARITY PINNING: partial. The reference holds four CRAM 3.0 files and no BAM twin of them; its own CRAM tests (src/cppNGS-TEST/BamReader_Test.cpp:400-560) need
the hg38 genome, which is not in the tree. Pinned here (tests/test_oracle_cram.py): every CRC of the fixtures, the record counts of containers against
slices, and the known answers of those tests that do not depend on the genome - name, position, end, CIGAR, qualities, mapping quality, insert size, mate
position, tags MC / AS of the first properly paired read, the CIGARs of the first two mapped reads and of the last read, the depths of the eight getPileup calls, the
allele split of the heterozygous SNPs (reads with a substitution feature at the site) and the insertion / deletion counts of the four indel pileups. BASES that come from the reference
genome cannot be checked on the fixtures (only those stored in features: soft clips, insertions, explicit bases).
Record by record, pure Python: fixture-sized inputs only.
"""
import bz2
import lzma
import struct
import zlib


class Cursor:
    def __init__(self, data, pos=0):
        self.d = data; self.p = pos

    def byte(self):
        v = self.d[self.p]; self.p += 1
        return v

    def take(self, n):
        v = self.d[self.p:self.p + n]
        if len(v) != n: raise ValueError("truncated CRAM data")
        self.p += n
        return v

    def i32(self):
        return struct.unpack("<i", self.take(4))[0]

    def u32(self):
        return struct.unpack("<I", self.take(4))[0]

    def itf8(self):
        b0 = self.byte()
        if b0 < 0x80: v = b0
        elif b0 < 0xc0: v = ((b0 & 0x3f) << 8) | self.byte()
        elif b0 < 0xe0: v = ((b0 & 0x1f) << 16) | (self.byte() << 8) | self.byte()
        elif b0 < 0xf0: v = ((b0 & 0x0f) << 24) | (self.byte() << 16) | (self.byte() << 8) | self.byte()
        else: v = ((b0 & 0x0f) << 28) | (self.byte() << 20) | (self.byte() << 12) | (self.byte() << 4) | (self.byte() & 0x0f)
        return v - (1 << 32) if v >= 1 << 31 else v

    def ltf8(self):
        b0 = self.byte(); n = 0
        while n < 8 and b0 & (0x80 >> n): n += 1
        if n == 8: v = 0
        else: v = b0 & (0xff >> (n + 1))
        for _ in range(n): v = (v << 8) | self.byte()
        return v - (1 << 64) if v >= 1 << 63 else v

    def array_itf8(self):
        return [self.itf8() for _ in range(self.itf8())]


# ------------------------------------------------------------------------------------------------------------------------------ rANS Nx16 (CRAM 3.1, block method 5)
# hts-specs CRAMcodecs "rANS Nx16" (htscodecs rANS_static4x16pr.c is the implementation htslib links; neither is in /root/reference and the reference holds
# no CRAM 3.1 file: this restatement is UNPINNED - it is held against oracle/cram_encode.py's writer of the same specification only).
# flags: 0x01 order-1, 0x04 32 interleaved states instead of 4, 0x08 striped, 0x10 no size, 0x20 stored, 0x40 run lengths, 0x80 bit packing
def _u7(c):
    v = 0
    while True:
        b = c.byte(); v = (v << 7) | (b & 0x7f)
        if not b & 0x80: return v


def _nx16_alphabet(c):
    A = [False] * 256; sym = c.byte(); last = sym; rle = 0
    while True:
        A[sym] = True
        if rle: rle -= 1; sym += 1
        else:
            sym = c.byte()
            if sym == last + 1: rle = c.byte()
        last = sym
        if sym == 0: break
    return A


def _nx16_scale(F, bits):
    tot = sum(F)
    if tot == 0 or tot == 1 << bits: return
    sh = 0
    while tot < 1 << bits: tot *= 2; sh += 1
    for i in range(256): F[i] <<= sh


def _nx16_lookup(F, bits):
    C = [0] * 257
    for i in range(256): C[i + 1] = C[i] + F[i]
    if C[256] > 1 << bits: raise ValueError("rANS Nx16 frequencies exceed the total")
    L = bytearray(1 << bits)
    for s in range(256):
        if F[s]: L[C[s]:C[s] + F[s]] = bytes([s]) * F[s]
    return C, L


def _nx16_order0(c, n, N):
    A = _nx16_alphabet(c); F = [0] * 256
    for s in range(256):
        if A[s]: F[s] = _u7(c)
    _nx16_scale(F, 12); C, L = _nx16_lookup(F, 12)
    R = [c.u32() for _ in range(N)]; out = bytearray(n)
    for i in range(n):
        j = i % N; f = R[j] & 0xfff; s = L[f]; out[i] = s
        x = F[s] * (R[j] >> 12) + f - C[s]
        if x < 1 << 15: x = (x << 16) | c.byte() | (c.byte() << 8)
        R[j] = x
    return bytes(out)


def _nx16_order1(c, n, N):
    comp = c.byte(); shift = comp >> 4
    tc = c
    if comp & 1:
        ulen = _u7(c); clen = _u7(c); tc = Cursor(_nx16_order0(Cursor(c.take(clen)), ulen, 4))
    A = _nx16_alphabet(tc); syms = [s for s in range(256) if A[s]]; T = {}
    for i in syms:
        F = [0] * 256; run = 0
        for j in syms:
            if run: run -= 1
            else:
                F[j] = _u7(tc)
                if F[j] == 0: run = tc.byte()
        _nx16_scale(F, shift); T[i] = (F,) + _nx16_lookup(F, shift)
    R = [c.u32() for _ in range(N)]; q = n // N; idx = [j * q for j in range(N)]; last = [0] * N; out = bytearray(n); mask = (1 << shift) - 1

    def step(j):
        F, C, L = T[last[j]]
        f = R[j] & mask; s = L[f]; out[idx[j]] = s; idx[j] += 1
        x = F[s] * (R[j] >> shift) + f - C[s]
        if x < 1 << 15: x = (x << 16) | c.byte() | (c.byte() << 8)
        R[j] = x; last[j] = s
    for _ in range(q):
        for j in range(N): step(j)
    while idx[N - 1] < n: step(N - 1)
    return bytes(out)


def rans_nx16_decode(c, n=None):
    """c: Cursor at the stream's flags byte; n: the decoded size when the caller knows it (a stream with the no-size flag)"""
    flags = c.byte()
    if not flags & 0x10: n = _u7(c)
    N = 32 if flags & 0x04 else 4
    if flags & 0x08:
        k = c.byte(); clen = [_u7(c) for _ in range(k)]; parts = []
        for j in range(k):
            sub = Cursor(c.take(clen[j])); parts.append(rans_nx16_decode(sub, n // k + (1 if n % k > j else 0)))
        out = bytearray(n)
        for j in range(k): out[j::k] = parts[j]
        return bytes(out)
    pack_len = rle_len = None
    if flags & 0x80:
        pack_len = n; nsym = c.byte(); P = [c.byte() for _ in range(nsym)]; n = _u7(c)
    if flags & 0x40:
        rle_len = n; mlen = _u7(c); n = _u7(c)
        if mlen & 1: meta = Cursor(c.take(mlen // 2))
        else:
            cm = _u7(c); meta = Cursor(_nx16_order0(Cursor(c.take(cm)), mlen // 2, 4))
        k = meta.byte() or 256; Lr = [False] * 256
        for _ in range(k): Lr[meta.byte()] = True
    if flags & 0x20: data = bytes(c.take(n))
    elif flags & 0x01: data = _nx16_order1(c, n, N)
    else: data = _nx16_order0(c, n, N)
    if flags & 0x40:
        out = bytearray()
        for s in data:
            if Lr[s]: out += bytes([s]) * (_u7(meta) + 1)
            else: out.append(s)
        if len(out) != rle_len: raise ValueError("rANS Nx16 run lengths do not add up")
        data = bytes(out)
    if flags & 0x80:
        if nsym <= 1: data = bytes([P[0]]) * pack_len if nsym else b""
        elif nsym <= 16:
            per, bits = (8, 1) if nsym <= 2 else ((4, 2) if nsym <= 4 else (2, 4))
            out = bytearray(pack_len)
            for i in range(pack_len): out[i] = P[(data[i // per] >> (bits * (i % per))) & ((1 << bits) - 1)]
            data = bytes(out)
    return data


# ------------------------------------------------------------------------------------------------------------------------------ rANS 4x8
def rans_decode(d):
    """CRAM 3.0 section 13 (rANS 4x8): byte order (0 / 1), uint32 compressed size, uint32 raw size, frequency table(s), 4 interleaved states"""
    order = d[0]; n_out = struct.unpack_from("<I", d, 5)[0]
    c = Cursor(d, 9)
    if n_out == 0: return b""

    def read_freqs():
        F = [0] * 256; sym = c.byte(); last = sym; rle = 0
        while True:
            f = c.byte()
            if f >= 0x80: f = ((f & 0x7f) << 8) | c.byte()
            F[sym] = f
            if rle: rle -= 1; sym += 1
            else:
                sym = c.byte()
                if sym == last + 1: rle = c.byte()
            last = sym
            if sym == 0: break
        C = [0] * 257
        for i in range(256): C[i + 1] = C[i] + F[i]
        lookup = bytearray(4096)
        for s in range(256):
            if F[s]: lookup[C[s]:C[s] + F[s]] = bytes([s]) * F[s]
        return F, C, lookup

    out = bytearray(n_out)
    if order == 0:
        F, C, L = read_freqs()
        R = [c.u32() for _ in range(4)]
        for i in range(n_out):
            j = i & 3; m = R[j] & 0xfff; s = L[m]; out[i] = s
            x = F[s] * (R[j] >> 12) + m - C[s]
            while x < (1 << 23): x = (x << 8) | c.byte()
            R[j] = x
        return bytes(out)
    tabs = {}
    ctx = c.byte(); last = ctx; rle = 0
    while True:
        tabs[ctx] = read_freqs()
        if rle: rle -= 1; ctx += 1
        else:
            ctx = c.byte()
            if ctx == last + 1: rle = c.byte()
        last = ctx
        if ctx == 0: break
    R = [c.u32() for _ in range(4)]
    q = n_out >> 2; idx = [0, q, 2 * q, 3 * q]; prev = [0, 0, 0, 0]

    def step(j):
        F, C, L = tabs[prev[j]]
        m = R[j] & 0xfff; s = L[m]; out[idx[j]] = s; idx[j] += 1
        x = F[s] * (R[j] >> 12) + m - C[s]
        while x < (1 << 23): x = (x << 8) | c.byte()
        R[j] = x; prev[j] = s
    for _ in range(q):
        for j in range(4): step(j)
    while idx[3] < n_out: step(3)
    return bytes(out)


# ------------------------------------------------------------------------------------------------------------------------------ structure
class Block:
    pass


def read_block(c):
    start = c.p; b = Block()
    b.method = c.byte(); b.ctype = c.byte(); b.cid = c.itf8(); csize = c.itf8(); b.rsize = c.itf8()
    raw = c.take(csize); crc = c.u32()
    if zlib.crc32(c.d[start:c.p - 4]) != crc: raise ValueError("CRAM block CRC mismatch")
    if b.method == 0: b.data = bytes(raw)
    elif b.method == 1: b.data = zlib.decompress(raw, 31)
    elif b.method == 2: b.data = bz2.decompress(raw)
    elif b.method == 3: b.data = lzma.decompress(raw)
    elif b.method == 4: b.data = rans_decode(raw)
    elif b.method == 5: b.data = rans_nx16_decode(Cursor(bytes(raw)))
    else: raise ValueError("CRAM block method %d (CRAM 3.1 codec: 6 arithmetic coder, 7 fqzcomp, 8 name tokeniser) is not supported" % b.method)
    if len(b.data) != b.rsize: raise ValueError("CRAM block inflates to another size than its header says")
    return b


class Container:
    pass


def read_container_header(c):
    start = c.p; k = Container()
    k.length = c.i32(); k.ref_id = c.itf8(); k.start = c.itf8(); k.span = c.itf8(); k.n_records = c.itf8(); k.counter = c.ltf8(); k.bases = c.ltf8()
    k.n_blocks = c.itf8(); k.landmarks = c.array_itf8(); crc = c.u32()
    if zlib.crc32(c.d[start:c.p - 4]) != crc: raise ValueError("CRAM container header CRC mismatch")
    return k


def read_encoding(c):
    codec = c.itf8(); n = c.itf8(); p = Cursor(c.take(n))
    if codec == 0: return ("NULL",)
    if codec == 1: return ("EXTERNAL", p.itf8())
    if codec == 3: return ("HUFFMAN", p.array_itf8(), p.array_itf8())
    if codec == 4: return ("BYTE_ARRAY_LEN", read_encoding(p), read_encoding(p))
    if codec == 5: return ("BYTE_ARRAY_STOP", p.byte(), p.itf8())
    if codec == 6: return ("BETA", p.itf8(), p.itf8())
    if codec == 7: return ("SUBEXP", p.itf8(), p.itf8())
    if codec == 9: return ("GAMMA", p.itf8())
    raise ValueError("CRAM encoding %d is not supported" % codec)


class CompressionHeader:
    pass


def read_compression_header(data):
    c = Cursor(data); h = CompressionHeader()
    h.RN = True; h.AP = True; h.RR = True; h.SM = None; h.TD = [[]]
    c.itf8()
    for _ in range(c.itf8()):
        key = bytes(c.take(2)).decode()
        if key in ("RN", "AP", "RR"): setattr(h, key, c.byte() != 0)
        elif key == "SM": h.SM = bytes(c.take(5))
        elif key == "TD":
            td = bytes(c.take(c.itf8())); h.TD = []
            for line in td.split(b"\0")[:-1] if td.endswith(b"\0") else td.split(b"\0"):
                h.TD.append([(line[i:i + 2], line[i + 2]) for i in range(0, len(line), 3)])
            if not h.TD: h.TD = [[]]
        else: raise ValueError("unknown preservation key " + key)
    h.ds = {}
    c.itf8()
    for _ in range(c.itf8()):
        key = bytes(c.take(2)).decode(); h.ds[key] = read_encoding(c)
    h.tags = {}
    c.itf8()
    for _ in range(c.itf8()):
        key = c.itf8(); h.tags[key] = read_encoding(c)
    # substitution matrix: for every reference base (ACGTN) the read bases in the order of their 2-bit codes
    h.subst = {}
    if h.SM:
        for i, ref in enumerate("ACGTN"):
            others = [b for b in "ACGTN" if b != ref]; byte = h.SM[i]; row = [None] * 4
            for k, b in enumerate(others): row[(byte >> (6 - 2 * k)) & 3] = b
            h.subst[ref] = row
    return h


class SliceHeader:
    pass


def read_slice_header(data):
    c = Cursor(data); s = SliceHeader()
    s.ref_id = c.itf8(); s.start = c.itf8(); s.span = c.itf8(); s.n_records = c.itf8(); s.counter = c.ltf8(); s.n_blocks = c.itf8()
    s.content_ids = c.array_itf8(); s.embedded_ref = c.itf8(); s.md5 = bytes(c.take(16))
    return s


# ------------------------------------------------------------------------------------------------------------------------------ decoders
class BitReader:
    def __init__(self, data):
        self.d = data; self.p = 0; self.bit = 7

    def bits(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | ((self.d[self.p] >> self.bit) & 1)
            self.bit -= 1
            if self.bit < 0: self.bit = 7; self.p += 1
        return v


class Decoders:
    """the decoders of one slice: the core bit stream and one cursor per external block"""
    def __init__(self, ch, core, ext):
        self.ch = ch; self.core = BitReader(core); self.ext = {k: Cursor(v) for k, v in ext.items()}
        self.huff = {}

    def _huffman(self, enc):
        key = id(enc)
        if key not in self.huff:
            syms, lens = enc[1], enc[2]
            order = sorted(range(len(syms)), key=lambda i: (lens[i], syms[i]))
            code = 0; last = 0; table = {}
            for i in order:
                code <<= (lens[i] - last); last = lens[i]
                table[(lens[i], code)] = syms[i]; code += 1
            self.huff[key] = (table, max(lens) if lens else 0, syms)
        return self.huff[key]

    def int(self, enc):
        kind = enc[0]
        if kind == "EXTERNAL": return self.ext[enc[1]].itf8()
        if kind == "HUFFMAN":
            table, mx, syms = self._huffman(enc)
            if mx == 0: return syms[0]
            code = 0
            for n in range(1, mx + 1):
                code = (code << 1) | self.core.bits(1)
                if (n, code) in table: return table[(n, code)]
            raise ValueError("bad Huffman code")
        if kind == "BETA": return self.core.bits(enc[2]) - enc[1]
        if kind == "GAMMA":
            n = 0
            while self.core.bits(1) == 0: n += 1
            return ((1 << n) | self.core.bits(n)) - enc[1]
        if kind == "SUBEXP":
            offset, k = enc[1], enc[2]; i = 0
            while self.core.bits(1) == 1: i += 1
            v = self.core.bits(k) if i == 0 else (1 << (i + k - 1)) | self.core.bits(i + k - 1)
            return v - offset
        raise ValueError("encoding %s cannot give an integer" % kind)

    def byte(self, enc):
        if enc[0] == "EXTERNAL": return self.ext[enc[1]].byte()
        return self.int(enc) & 0xff

    def bytes_n(self, enc, n):
        if enc[0] == "EXTERNAL": return bytes(self.ext[enc[1]].take(n))
        return bytes(self.byte(enc) for _ in range(n))

    def array(self, enc):
        if enc[0] == "BYTE_ARRAY_STOP":
            c = self.ext[enc[2]]; e = c.d.index(bytes([enc[1]]), c.p); v = bytes(c.d[c.p:e]); c.p = e + 1
            return v
        if enc[0] == "BYTE_ARRAY_LEN":
            n = self.int(enc[1]); return self.bytes_n(enc[2], n)
        raise ValueError("encoding %s cannot give a byte array" % enc[0])


# ------------------------------------------------------------------------------------------------------------------------------ records
class Rec:
    pass


BAM_FPAIRED, BAM_FPROPER, BAM_FUNMAP, BAM_FMUNMAP, BAM_FREVERSE, BAM_FMREVERSE, BAM_FREAD1 = 1, 2, 4, 8, 16, 32, 64
CF_QUAL_ARRAY, CF_DETACHED, CF_MATE_DOWNSTREAM, CF_NO_SEQ = 1, 2, 4, 8


def decode_slice(ch, sh, blocks, ref_fetch):
    """-> [Rec]. ref_fetch(ref_id, start0, length) -> bytes of the reference (upper case) or None when no genome is at hand (those bases come out as 'N'
    and Rec.bases_from_ref is set)"""
    core = next(b.data for b in blocks if b.ctype == 5)
    ext = {b.cid: b.data for b in blocks if b.ctype == 4}
    D = Decoders(ch, core, ext); ds = ch.ds
    recs = []; prev_pos = sh.start
    embedded = ext.get(sh.embedded_ref) if sh.embedded_ref >= 0 else None
    for i in range(sh.n_records):
        r = Rec(); r.bf = D.int(ds["BF"]); r.cf = D.int(ds["CF"])
        r.ref_id = D.int(ds["RI"]) if sh.ref_id == -2 else sh.ref_id
        r.rl = D.int(ds["RL"])
        ap = D.int(ds["AP"])
        if ch.AP: prev_pos += ap; r.pos = prev_pos          # 1-based
        else: r.pos = ap
        r.rg = D.int(ds["RG"])
        r.name = D.array(ds["RN"]) if ch.RN else None
        r.mate_line = -1; r.mf = 0; r.ns = -1; r.np = 0; r.ts = 0; r.nf = None
        if r.cf & CF_DETACHED:
            r.mf = D.int(ds["MF"])
            if not ch.RN: r.name = D.array(ds["RN"])
            r.ns = D.int(ds["NS"]); r.np = D.int(ds["NP"]); r.ts = D.int(ds["TS"])
        elif r.cf & CF_MATE_DOWNSTREAM:
            r.nf = D.int(ds["NF"]); r.mate_line = i + r.nf + 1
        tl = D.int(ds["TL"]); r.tags = []
        for tag, typ in ch.TD[tl]:
            key = (tag[0] << 16) | (tag[1] << 8) | typ
            r.tags.append((bytes(tag), typ, D.array(ch.tags[key])))
        r.features = []; r.mapq = 0; r.bases_from_ref = False
        if not r.bf & BAM_FUNMAP:
            fn = D.int(ds["FN"]); fpos = 0
            for _ in range(fn):
                code = chr(D.byte(ds["FC"])); fpos += D.int(ds["FP"])
                if code == "B": v = (D.byte(ds["BA"]), D.byte(ds["QS"]))
                elif code == "X": v = D.byte(ds["BS"])
                elif code == "I": v = D.array(ds["IN"])
                elif code == "S": v = D.array(ds["SC"])
                elif code == "H": v = D.int(ds["HC"])
                elif code == "P": v = D.int(ds["PD"])
                elif code == "D": v = D.int(ds["DL"])
                elif code == "N": v = D.int(ds["RS"])
                elif code == "i": v = D.byte(ds["BA"])
                elif code == "b": v = D.array(ds["BB"])
                elif code == "q": v = D.array(ds["QQ"])
                elif code == "Q": v = D.byte(ds["QS"])
                else: raise ValueError("unknown read feature " + code)
                r.features.append((code, fpos, v))
            r.mapq = D.int(ds["MQ"])
            r.qual = D.bytes_n(ds["QS"], r.rl) if r.cf & CF_QUAL_ARRAY else None
            build_alignment(ch, sh, r, ref_fetch, embedded)
        else:
            r.seq = b"*" if r.cf & CF_NO_SEQ else D.bytes_n(ds["BA"], r.rl)
            r.qual = D.bytes_n(ds["QS"], r.rl) if r.cf & CF_QUAL_ARRAY else None
            r.cigar = []; r.end = r.pos
            if r.cf & CF_NO_SEQ: r.seq = None
        recs.append(r)
    resolve_mates(recs)
    return recs


def build_alignment(ch, sh, r, ref_fetch, embedded):
    """read features -> CIGAR, bases, (qualities): CRAMv3 section 10.6"""
    cig = []

    def add(op, n):
        if n <= 0: return
        if cig and cig[-1][0] == op: cig[-1][1] += n
        else: cig.append([op, n])
    seq = bytearray(b"N" * r.rl); qual = bytearray(r.qual) if r.qual is not None else bytearray(b"\xff" * r.rl)
    ref_pos = r.pos - 1      # 0-based on the reference
    read_pos = 0             # 0-based in the read
    no_ref = [False]

    def ref_bases(p0, n):
        if n <= 0: return b""
        if embedded is not None:
            o = p0 - (sh.start - 1); return embedded[o:o + n].upper()
        got = ref_fetch(r.ref_id, p0, n) if ref_fetch else None
        if got is None: no_ref[0] = True; return b"N" * n
        return got.upper() + b"N" * (n - len(got))

    def match_to(upto):     # bases between features come from the reference
        nonlocal ref_pos, read_pos
        n = upto - read_pos
        if n > 0:
            seq[read_pos:upto] = ref_bases(ref_pos, n); add("M", n); ref_pos += n; read_pos = upto
    for code, fpos, v in r.features:
        at = fpos - 1
        if code in "qQ":       # (htslib reads the quality array behind the features: where one is stored it overwrites what the features set)
            if r.qual is None:
                if code == "Q": qual[at] = v
                else: qual[at:at + len(v)] = v
            continue
        match_to(at)
        if code == "B":
            seq[at] = v[0]; add("M", 1); ref_pos += 1; read_pos += 1
            if r.qual is None: qual[at] = v[1]
        elif code == "X":
            rb = ref_bases(ref_pos, 1).decode()
            if rb not in "ACGTN": rb = "N"
            seq[at] = ord(ch.subst[rb][v]); add("M", 1); ref_pos += 1; read_pos += 1
        elif code == "I": seq[at:at + len(v)] = v; add("I", len(v)); read_pos += len(v)
        elif code == "i": seq[at] = v; add("I", 1); read_pos += 1
        elif code == "S": seq[at:at + len(v)] = v; add("S", len(v)); read_pos += len(v)
        elif code == "b": seq[at:at + len(v)] = v; add("M", len(v)); ref_pos += len(v); read_pos += len(v)
        elif code == "D": add("D", v); ref_pos += v
        elif code == "N": add("N", v); ref_pos += v
        elif code == "H": add("H", v)
        elif code == "P": add("P", v)
    match_to(r.rl)
    r.cigar = [(op, n) for op, n in cig]; r.end = ref_pos          # 1-based closed end = 0-based end
    if r.cf & CF_NO_SEQ: r.seq = None
    else: r.seq = bytes(seq)
    r.qual = bytes(qual) if r.qual is not None or any(f[0] in "BqQ" for f in r.features) else None
    r.bases_from_ref = no_ref[0]
    if not cig: r.end = r.pos


def resolve_mates(recs):
    """mate fields of the records of one slice (htslib cram_decode.c cram_decode_slice_xref): detached records carry them; the members of a chain linked by NF
    take them from each other, and the template length is computed over the chain"""
    n = len(recs)
    for r in recs:
        r.mate_ref = -1; r.mate_pos = 0; r.tlen = None
    for i, r in enumerate(recs):
        if r.cf & CF_DETACHED:
            r.mate_ref = r.ns; r.mate_pos = r.np; r.tlen = r.ts
            if r.mf & 1: r.bf |= BAM_FMREVERSE
            if r.mf & 2: r.bf |= BAM_FMUNMAP
            continue
        if r.mate_line < 0: continue
        if r.mate_line >= n: raise ValueError("mate chain leaves the slice")
        if r.tlen is None:
            # the first member of a chain: leftmost start, rightmost end, how many members start leftmost, whether all sit on one reference
            left = r.pos; right = r.end; left_cnt = 0; ref = r.ref_id; j = i; chain = []
            while True:
                m = recs[j]; chain.append(j)
                if m.pos < left: left = m.pos; left_cnt = 1
                elif m.pos == left: left_cnt += 1
                if m.end > right: right = m.end
                if m.ref_id != ref: ref = -1
                if m.mate_line == -1:
                    m.mate_line = i; break          # the last member points back at the first
                if m.mate_line <= j or m.mate_line >= n: raise ValueError("bad mate chain")
                j = m.mate_line
            tlen = right - left + 1
            for j in chain:
                m = recs[j]
                if ref == -1: m.tlen = 0
                elif m.pos == left and (left_cnt == 1 or m.bf & BAM_FREAD1): m.tlen = tlen    # ties: the first read of the template takes the positive length
                else: m.tlen = -tlen
        mate = recs[r.mate_line]
        r.mate_ref = mate.ref_id; r.mate_pos = mate.pos
        r.bf |= BAM_FPAIRED
        if mate.bf & BAM_FUNMAP: r.bf |= BAM_FMUNMAP; r.tlen = 0
        if r.bf & BAM_FUNMAP: r.tlen = 0
        if mate.bf & BAM_FREVERSE: r.bf |= BAM_FMREVERSE
    for r in recs:
        if r.tlen is None: r.tlen = 0


# ------------------------------------------------------------------------------------------------------------------------------ whole file
class CramFile:
    pass


def read_cram(path, ref_fetch=None):
    """-> CramFile: .version, .header (SAM text), .containers [(container header, [slice headers])], .records [Rec] in file order"""
    d = open(path, "rb").read()
    if d[:4] != b"CRAM": raise ValueError("not a CRAM file")
    f = CramFile(); f.version = (d[4], d[5])
    if d[4] != 3 or d[5] not in (0, 1): raise ValueError("CRAM %d.%d is not supported (3.0 and 3.1 only)" % f.version)
    c = Cursor(d, 26)
    k = read_container_header(c); end = c.p + k.length
    b = read_block(c)
    if b.ctype != 0: raise ValueError("the first container does not hold the SAM header")
    l_text = struct.unpack_from("<i", b.data, 0)[0]; f.header = b.data[4:4 + l_text].decode()
    c.p = end
    f.containers = []; f.records = []; f.eof = False
    while c.p < len(d):
        k = read_container_header(c); end = c.p + k.length
        if k.n_records == 0 and k.ref_id == -1 and k.start == 4542278:   # the EOF container
            f.eof = True; c.p = end; continue
        ch_block = read_block(c)
        if ch_block.ctype != 1: raise ValueError("compression header expected")
        ch = read_compression_header(ch_block.data); slices = []
        while c.p < end:
            sb = read_block(c)
            if sb.ctype != 2: raise ValueError("slice header expected")
            sh = read_slice_header(sb.data)
            blocks = [read_block(c) for _ in range(sh.n_blocks)]
            f.records.extend(decode_slice(ch, sh, blocks, ref_fetch)); slices.append(sh)
        f.containers.append((k, slices, ch))
    return f


def ref_names(header):
    out = []
    for line in header.split("\n"):
        if line.startswith("@SQ"):
            fields = dict(x.split(":", 1) for x in line.split("\t")[1:])
            out.append((fields["SN"], int(fields["LN"])))
    return out


def read_groups(header):
    return [dict(x.split(":", 1) for x in line.split("\t")[1:])["ID"] for line in header.split("\n") if line.startswith("@RG")]


CIGAR_OPS = "MIDNSHP=X"


def to_bam_record(r, rgs):
    """one BAM record (SAM spec 4.2) as htslib's cram_to_bam builds it: the tags of the record, then RG from the read-group index"""
    name = (r.name or b"*") + b"\0"
    cig = b"".join(struct.pack("<I", n << 4 | CIGAR_OPS.index(op)) for op, n in r.cigar)
    seq = r.seq if r.seq is not None else b""
    l_seq = len(seq)
    packed = bytearray((l_seq + 1) // 2)
    code = {c: i for i, c in enumerate(b"=ACMGRSVTWYHKDBN")}
    for i, ch in enumerate(seq):
        v = code.get(ch, 15)
        packed[i >> 1] |= v << 4 if not i & 1 else v
    qual = r.qual if (r.qual is not None and l_seq) else b"\xff" * l_seq
    tags = b""
    for tag, typ, val in r.tags: tags += tag + bytes([typ]) + val
    if r.rg >= 0 and r.rg < len(rgs): tags += b"RGZ" + rgs[r.rg].encode() + b"\0"
    pos0 = r.pos - 1
    end0 = r.end if r.cigar else pos0 + 1
    bin_ = reg2bin(pos0, end0) if r.ref_id >= 0 or pos0 >= 0 else 4680
    if pos0 < 0: bin_ = 4680
    n_cig = len(r.cigar)
    if n_cig > 65535:
        # SAM spec 4.2.2 / htslib bam_write1: n_cigar_op is 16 bits - the record carries the placeholder <l_seq>S<reference length>N and its operations in a
        # CG:B,I tag behind all other tags (readers put them back: htslib bam_tag2cigar, oracle/bamio.hpp, K2 / K3)
        tags += b"CGBI" + struct.pack("<i", n_cig) + cig
        cig = struct.pack("<II", l_seq << 4 | 4, (end0 - pos0) << 4 | 3); n_cig = 2
    body = struct.pack("<iiBBHHHiiii", r.ref_id, pos0, len(name), r.mapq, bin_, n_cig, r.bf, l_seq, r.mate_ref, r.mate_pos - 1, r.tlen) + name + cig + bytes(packed) + qual[:l_seq] + tags
    return struct.pack("<i", len(body)) + body


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14: return 4681 + (beg >> 14)
    if beg >> 17 == end >> 17: return 585 + (beg >> 17)
    if beg >> 20 == end >> 20: return 73 + (beg >> 20)
    if beg >> 23 == end >> 23: return 9 + (beg >> 23)
    if beg >> 26 == end >> 26: return 1 + (beg >> 26)
    return 0


def cigar_string(r):
    return "".join("%d%s" % (n, op) for op, n in r.cigar) or "*"
