"""ORACLE (test infrastructure, never imported by the product): a CRAM 3.0 WRITER, so that the product's CRAM input can be tested against BAM truth.

The reference holds no BAM twin of its CRAM fixtures and no genome for them, so parity on reference-derived bases, on the slices' MD5 check and on
tool outputs cannot come from the fixtures alone. This writer turns a BAM (+ a genome) into a CRAM 3.0 file following hts-specs CRAMv3; the product must then
return exactly the BAM's records. What it writes is deliberately varied so that every decoder path runs:
  * blocks: raw, gzip, rANS 4x8 order 0 and order 1 (the encoder of section 13 is here as well)
  * encodings: EXTERNAL, HUFFMAN (one symbol and several), BETA, GAMMA, SUBEXP, BYTE_ARRAY_STOP, BYTE_ARRAY_LEN
  * single-reference slices with delta positions and MD5, multi-reference slices (RI series), unmapped slices, an embedded reference, RR = false ('b' features)
  * mates: chains inside a slice (NF) where the decoder's rules reproduce the BAM's fields (as htslib's writer decides), detached otherwise
The writer is checked two ways (tests/test_oracle_cram.py): oracle/cram_decode.py - pinned on the reference's htslib-written fixtures - reads its files back to
the BAM's records, and its EOF container equals the fixtures' byte for byte.
"""
import hashlib
import struct
import zlib

import cram_decode as CD

BASES = "ACGTN"
NAME_METHOD = None   # (encode(name_method=...): the block method of the read names' external block)


def itf8(v):
    v &= 0xffffffff
    if v < 0x80: return bytes([v])
    if v < 0x4000: return bytes([0x80 | (v >> 8), v & 0xff])
    if v < 0x200000: return bytes([0xc0 | (v >> 16), (v >> 8) & 0xff, v & 0xff])
    if v < 0x10000000: return bytes([0xe0 | (v >> 24), (v >> 16) & 0xff, (v >> 8) & 0xff, v & 0xff])
    return bytes([0xf0 | (v >> 28), (v >> 20) & 0xff, (v >> 12) & 0xff, (v >> 4) & 0xff, v & 0x0f])


def ltf8(v):
    v &= (1 << 64) - 1
    for n in range(8):
        if v < 1 << (7 * (n + 1)):
            first = ((0xff << (8 - n)) & 0xff) | (v >> (8 * n))
            return bytes([first]) + bytes((v >> (8 * k)) & 0xff for k in range(n - 1, -1, -1))
    return b"\xff" + v.to_bytes(8, "big")


def array_itf8(a):
    return itf8(len(a)) + b"".join(itf8(x) for x in a)


# ------------------------------------------------------------------------------------------------------------------------------ rANS 4x8 encoder
def _normalise(counts):
    total = sum(counts)
    F = [0] * 256
    if total == 0: return F
    for s in range(256):
        if counts[s]: F[s] = max(1, counts[s] * 4096 // total)
    diff = 4096 - sum(F)
    while diff != 0:
        s = max(range(256), key=lambda x: F[x])
        step = diff if diff > 0 else max(diff, -(F[s] - 1))
        if step == 0: raise ValueError("cannot normalise frequencies")
        F[s] += step; diff -= step
    return F


def _write_freqs(F):
    out = bytearray(); rle = 0
    for j in range(256):
        if not F[j]: continue
        if rle: rle -= 1
        else:
            out.append(j)
            if j and F[j - 1]:
                r = j + 1
                while r < 256 and F[r]: r += 1
                rle = r - (j + 1); out.append(rle)
        if F[j] < 128: out.append(F[j])
        else: out += bytes([128 | (F[j] >> 8), F[j] & 0xff])
    out.append(0)
    return bytes(out)


def rans_encode(data, order):
    n = len(data)
    head = lambda body: bytes([order]) + struct.pack("<II", len(body), n) + body
    if n == 0: return head(b"")
    q = n >> 2
    # the decoder's symbol events in forward order: (state, context, symbol)
    events = []
    if order == 0:
        counts = [0] * 256
        for b in data: counts[b] += 1
        F = {0: _normalise(counts)}
        table = _write_freqs(F[0])
        events = [(i & 3, 0, data[i]) for i in range(n)]
    else:
        idx = [0, q, 2 * q, 3 * q]; prev = [0, 0, 0, 0]
        for _ in range(q):
            for j in range(4):
                s = data[idx[j]]; events.append((j, prev[j], s)); prev[j] = s; idx[j] += 1
        while idx[3] < n:
            s = data[idx[3]]; events.append((3, prev[3], s)); prev[3] = s; idx[3] += 1
        counts = {}
        for _, c, s in events: counts.setdefault(c, [0] * 256)[s] += 1
        F = {c: _normalise(v) for c, v in counts.items()}
        present = [1 if c in F else 0 for c in range(256)]
        tab = bytearray(); rle = 0
        for c in range(256):
            if not present[c]: continue
            if rle: rle -= 1
            else:
                tab.append(c)
                if c and present[c - 1]:
                    r = c + 1
                    while r < 256 and present[r]: r += 1
                    rle = r - (c + 1); tab.append(rle)
            tab += _write_freqs(F[c])
        tab.append(0); table = bytes(tab)
    C = {}
    for c, f in F.items():
        acc = 0; cc = [0] * 256
        for s in range(256): cc[s] = acc; acc += f[s]
        C[c] = cc
    R = [1 << 23] * 4; buf = bytearray()     # bytes in emission order; the stream is their reverse
    for j, c, s in reversed(events):
        f = F[c][s]; x = R[j]; x_max = (1 << 19) * f
        while x >= x_max: buf.append(x & 0xff); x >>= 8
        R[j] = ((x // f) << 12) + (x % f) + C[c][s]
    for j in (3, 2, 1, 0):
        x = R[j]; buf += bytes([(x >> 24) & 0xff, (x >> 16) & 0xff, (x >> 8) & 0xff, x & 0xff])
    return head(table + bytes(reversed(buf)))


# ------------------------------------------------------------------------------------------------------------------------------ rANS Nx16 (CRAM 3.1)
# The writer's side of hts-specs CRAMcodecs "rANS Nx16" (see oracle/cram_decode.py rans_nx16_decode: unpinned - no CRAM 3.1 file in the reference).
def u7(v):
    out = [v & 0x7f]; v >>= 7
    while v: out.append((v & 0x7f) | 0x80); v >>= 7
    return bytes(reversed(out))


def _nx16_alphabet(present):
    tab = bytearray(); rle = 0
    for c in range(256):
        if not present[c]: continue
        if rle: rle -= 1
        else:
            tab.append(c)
            if c and present[c - 1]:
                r = c + 1
                while r < 256 and present[r]: r += 1
                rle = r - (c + 1); tab.append(rle)
    tab.append(0)
    return bytes(tab)


def _scale_to(counts, total):
    """frequencies that add up to total exactly, every counted symbol at least 1"""
    n = sum(counts); F = [0] * 256
    if n == 0: return F
    for s in range(256):
        if counts[s]: F[s] = max(1, counts[s] * total // n)
    d = total - sum(F); order = sorted(range(256), key=lambda s: -F[s])
    k = 0
    while d != 0:
        s = order[k % 256]
        if d > 0: F[s] += 1; d -= 1
        elif F[s] > 1: F[s] -= 1; d += 1
        k += 1
    return F


def _nx16_states(events, F, C, shift, N):
    """events in decoding order: (state, context, symbol) -> N final states + the 16-bit renormalisation words in reading order"""
    R = [1 << 15] * N; words = []
    for j, c, s in reversed(events):
        f = F[c][s]; x = R[j]
        if x >= ((1 << 15) >> shift << 16) * f: words.append(x & 0xffff); x >>= 16
        R[j] = ((x // f) << shift) + (x % f) + C[c][s]
    return b"".join(struct.pack("<I", x) for x in R) + b"".join(struct.pack("<H", w) for w in reversed(words))


def _nx16_order0(data, N, half_total=False):
    counts = [0] * 256
    for b in data: counts[b] += 1
    F = _scale_to(counts, 4096)
    stored = F
    if half_total and all(f % 2 == 0 for f in F): stored = [f // 2 for f in F]   # (a table that adds up to a smaller power of two: the reader scales it up)
    acc = 0; C = [0] * 256
    for s in range(256): C[s] = acc; acc += F[s]
    table = _nx16_alphabet([f > 0 for f in F]) + b"".join(u7(stored[s]) for s in range(256) if F[s])
    return table + _nx16_states([(i % N, 0, data[i]) for i in range(len(data))], {0: F}, {0: C}, 12, N)


def _nx16_order1(data, N, shift=12, comp_table=False):
    n = len(data); q = n // N; events = []; idx = [j * q for j in range(N)]; last = [0] * N
    for _ in range(q):
        for j in range(N):
            s = data[idx[j]]; events.append((j, last[j], s)); last[j] = s; idx[j] += 1
    while idx[N - 1] < n:
        s = data[idx[N - 1]]; events.append((N - 1, last[N - 1], s)); last[N - 1] = s; idx[N - 1] += 1
    counts = {}
    for _, c, s in events: counts.setdefault(c, [0] * 256)[s] += 1
    present = [False] * 256
    for c in counts: present[c] = True
    for b in data: present[b] = True
    syms = [s for s in range(256) if present[s]]
    F = {c: _scale_to(counts.get(c, [0] * 256), 1 << shift) for c in syms}; C = {}
    for c in syms:
        acc = 0; cc = [0] * 256
        for s in range(256): cc[s] = acc; acc += F[c][s]
        C[c] = cc
    tab = bytearray(_nx16_alphabet(present))
    for i in syms:
        k = 0
        while k < len(syms):
            f = F[i][syms[k]]; tab += u7(f); k += 1
            if f == 0:
                run = 0
                while k < len(syms) and F[i][syms[k]] == 0 and run < 255: run += 1; k += 1
                tab.append(run)
    body = _nx16_states(events, F, C, shift, N)
    if comp_table:
        ct = _nx16_order0(bytes(tab), 4)
        return bytes([(shift << 4) | 1]) + u7(len(tab)) + u7(len(ct)) + ct + body
    return bytes([shift << 4]) + bytes(tab) + body


def rans_nx16_encode(data, order=0, x32=False, pack=False, rle=False, stripe=0, cat=False, nosz=False, comp_table=False, shift=12, comp_meta=False, half_total=False):
    data = bytes(data); n = len(data); N = 32 if x32 else 4
    flags = (1 if order else 0) | (4 if x32 else 0) | (0x10 if nosz else 0)
    head = b"" if nosz else u7(n)
    if n == 0: return bytes([flags | 0x20]) + head
    if stripe:
        parts = [rans_nx16_encode(data[j::stripe], order=order, x32=x32, nosz=True) for j in range(stripe)]
        return bytes([flags | 0x08]) + head + bytes([stripe]) + b"".join(u7(len(p)) for p in parts) + b"".join(parts)
    meta = b""
    if pack:
        syms = sorted(set(data)); nsym = len(syms)
        if nsym <= 16:
            flags |= 0x80
            if nsym <= 1: packed = b""
            else:
                per, bits = (8, 1) if nsym <= 2 else ((4, 2) if nsym <= 4 else (2, 4))
                code = {s: k for k, s in enumerate(syms)}; packed = bytearray((n + per - 1) // per)
                for i, b in enumerate(data): packed[i // per] |= code[b] << (bits * (i % per))
            meta += bytes([nsym]) + bytes(syms) + u7(len(packed)); data = bytes(packed); n = len(data)
    if rle and n:
        # symbols whose runs are worth their length bytes: those that repeat at all
        runs = []; i = 0
        while i < n:
            j = i
            while j + 1 < n and data[j + 1] == data[i]: j += 1
            runs.append((data[i], j - i)); i = j + 1
        rl = sorted({s for s, k in runs if k > 0})
        if rl:
            flags |= 0x40
            lit = bytearray(); lens = bytearray()
            for s, k in runs:
                if s in rl: lit.append(s); lens += u7(k)
                else: lit += bytes([s]) * (k + 1)
            rmeta = bytes([len(rl) & 255]) + bytes(rl) + bytes(lens)
            if comp_meta:
                cm = _nx16_order0(rmeta, 4)
                meta += u7(len(rmeta) * 2) + u7(len(lit)) + u7(len(cm)) + cm
            else: meta += u7(len(rmeta) * 2 + 1) + u7(len(lit)) + rmeta
            data = bytes(lit); n = len(data)
    if cat or n == 0 or (flags & 0x80 and n == 0): return bytes([flags | 0x20]) + head + meta + data
    if order and n >= N: body = _nx16_order1(data, N, shift, comp_table)
    else: flags &= ~1; body = _nx16_order0(data, N, half_total)
    return bytes([flags]) + head + meta + body


# ------------------------------------------------------------------------------------------------------------------------------ blocks
def block(method, ctype, cid, data):
    if method == 0: comp = data
    elif method == 1:
        c = zlib.compressobj(6, zlib.DEFLATED, 31); comp = c.compress(data) + c.flush()
    elif method == 2:
        import bz2; comp = bz2.compress(data)
    elif method == 3:
        import lzma; comp = lzma.compress(data)
    elif method == 4: comp = rans_encode(data, 0)
    elif method == 41: comp = rans_encode(data, 1); method = 4
    elif 50 <= method <= 59:
        # CRAM 3.1 rANS Nx16, one shape per code: 50 order 0 | 51 order 1 | 52 order 0, 32 states | 53 order 1, 32 states, compressed table, 10-bit frequencies |
        # 54 bit packing + order 0 | 55 run lengths + order 1 | 56 four stripes | 57 stored | 58 run lengths (compressed) + packing | 59 order 0, table adds up to 2048
        kw = {50: {}, 51: dict(order=1), 52: dict(x32=True), 53: dict(order=1, x32=True, comp_table=True, shift=10), 54: dict(pack=True), 55: dict(order=1, rle=True),
              56: dict(stripe=4), 57: dict(cat=True), 58: dict(rle=True, pack=True, comp_meta=True), 59: dict(half_total=True)}[method]
        comp = rans_nx16_encode(data, **kw); method = 5
    elif method == 80: comp = zlib.compress(data); method = 8   # (a block of the name tokeniser's method number: this writer has no such codec - for readers that must refuse it, or never look)
    else: raise ValueError("block method")
    b = bytes([method, ctype]) + itf8(cid) + itf8(len(comp)) + itf8(len(data)) + comp
    return b + struct.pack("<I", zlib.crc32(b))


def container(ref_id, start, span, n_records, counter, bases, blocks, landmarks):
    body = b"".join(blocks)
    h = struct.pack("<i", len(body)) + itf8(ref_id) + itf8(start) + itf8(span) + itf8(n_records) + ltf8(counter) + ltf8(bases) + itf8(len(blocks)) + array_itf8(landmarks)
    return h + struct.pack("<I", zlib.crc32(h)) + body


def eof_container():
    return container(-1, 4542278, 0, 0, 0, 0, [block(0, 1, 0, b"\x01\x00\x01\x00\x01\x00")], [])


# ------------------------------------------------------------------------------------------------------------------------------ BAM side
def read_bam(path):
    """-> (header text, [(name, length)], [record dict])"""
    img = open(path, "rb").read(); pos = 0; s = bytearray()
    while pos < len(img):
        bs = struct.unpack_from("<H", img, pos + 16)[0] + 1
        s += zlib.decompress(img[pos + 18:pos + bs - 8], -15); pos += bs
    l_text = struct.unpack_from("<i", s, 4)[0]; text = bytes(s[8:8 + l_text]).decode(); o = 8 + l_text
    n_ref = struct.unpack_from("<i", s, o)[0]; o += 4; refs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", s, o)[0]; refs.append((bytes(s[o + 4:o + 3 + ln]).decode(), struct.unpack_from("<i", s, o + 4 + ln)[0])); o += 8 + ln
    recs = []
    while o < len(s):
        recs.append(parse_record(bytes(s[o:o + 4 + struct.unpack_from("<i", s, o)[0]]))); o += len(recs[-1]["raw"])
    return text, refs, recs


def parse_record(raw):
    bs, ref_id, pos0, l_name, mapq, bin_, n_cig, flag, l_seq, mref, mpos0, tlen = struct.unpack_from("<iiiBBHHHiiii", raw, 0)
    o = 36; name = raw[o:o + l_name - 1]; o += l_name
    cigar = [(CD.CIGAR_OPS[c & 15], c >> 4) for c in struct.unpack_from("<%dI" % n_cig, raw, o)]; o += 4 * n_cig
    packed = raw[o:o + (l_seq + 1) // 2]; o += (l_seq + 1) // 2
    seq = bytes(b"=ACMGRSVTWYHKDBN"[(packed[i >> 1] >> (4 if not i & 1 else 0)) & 15] for i in range(l_seq))
    qual = raw[o:o + l_seq]; o += l_seq
    tags = split_tags(raw[o:])
    # htslib bam_tag2cigar (sam_read1 of a BAM): a placeholder CIGAR <l_seq>S... with a CG:B,I tag of at least as many operations stands for a CIGAR of more
    # than 65535 operations - the reader puts it back and drops the tag; the CRAM then holds the real operations as read features
    if n_cig and ref_id >= 0 and pos0 >= 0 and cigar[0] == ("S", l_seq):
        cg = [t for t in tags if t[0] == b"CG" and t[1] == ord("B") and t[2][:1] == b"I"]
        if cg:
            n = struct.unpack_from("<i", cg[0][2], 1)[0]
            if n_cig <= n < 1 << 29:
                cigar = [(CD.CIGAR_OPS[c & 15], c >> 4) for c in struct.unpack_from("<%dI" % n, cg[0][2], 5)]
                tags = [t for t in tags if t is not cg[0]]
    return dict(raw=raw, ref_id=ref_id, pos=pos0 + 1, mapq=mapq, flag=flag, mate_ref=mref, mate_pos=mpos0 + 1, tlen=tlen, name=name, cigar=cigar, seq=seq, qual=qual, tags=tags)


def split_tags(aux):
    out = []; o = 0
    size = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}
    while o < len(aux):
        tag = aux[o:o + 2]; typ = chr(aux[o + 2]); v0 = o + 3
        if typ in size: v1 = v0 + size[typ]
        elif typ in "ZH": v1 = aux.index(b"\0", v0) + 1
        elif typ == "B":
            sub = chr(aux[v0]); cnt = struct.unpack_from("<i", aux, v0 + 1)[0]; v1 = v0 + 5 + cnt * size[sub]
        else: raise ValueError("aux type " + typ)
        out.append((bytes(tag), ord(typ), bytes(aux[v0:v1]))); o = v1
    return out


def ref_end(r):
    n = sum(k for op, k in r["cigar"] if op in "MDN=X")
    return r["pos"] + n - 1 if r["cigar"] and not r["flag"] & 4 else r["pos"]


# ------------------------------------------------------------------------------------------------------------------------------ the writer
SUBST_DEFAULT = bytes([0x1b, 0x1b, 0x1b, 0x1b, 0x1b])      # every row: the other bases in alphabetical order take codes 0, 1, 2, 3


class SliceWriter:
    """collects the data series of one slice: external blocks by content id and the core bit stream"""
    def __init__(self, enc):
        self.enc = enc; self.ext = {}; self.bits = []

    def _bits(self, v, n):
        for k in range(n - 1, -1, -1): self.bits.append((v >> k) & 1)

    def put_int(self, key, v, enc=None):
        e = enc or self.enc[key]; kind = e[0]
        if kind == "EXTERNAL": self.ext.setdefault(e[1], bytearray()).extend(itf8(v))
        elif kind == "HUFFMAN":
            syms, lens = e[1], e[2]
            if max(lens) == 0:
                assert v == syms[0], (key, v, syms); return
            order = sorted(range(len(syms)), key=lambda i: (lens[i], syms[i])); code = 0; last = 0
            for i in order:
                code <<= (lens[i] - last); last = lens[i]
                if syms[i] == v: self._bits(code, lens[i]); return
                code += 1
            raise ValueError("value without a Huffman code: %s %d" % (key, v))
        elif kind == "BETA": self._bits(v + e[1], e[2])
        elif kind == "GAMMA":
            x = v + e[1]; assert x > 0
            n = x.bit_length() - 1; self._bits(0, n); self._bits(1, 1); self._bits(x & ((1 << n) - 1), n)
        elif kind == "SUBEXP":
            x = v + e[1]; k = e[2]; assert x >= 0
            if x < (1 << k): self._bits(0, 1); self._bits(x, k)
            else:
                b = x.bit_length() - 1; i = b - k + 1
                self._bits((1 << i) - 1, i); self._bits(0, 1); self._bits(x & ((1 << b) - 1), b)
        else: raise ValueError(kind)

    def put_byte(self, key, v):
        e = self.enc[key]
        if e[0] == "EXTERNAL": self.ext.setdefault(e[1], bytearray()).append(v)
        else: self.put_int(key, v)

    def put_bytes(self, key, data):
        e = self.enc[key]; assert e[0] == "EXTERNAL"
        self.ext.setdefault(e[1], bytearray()).extend(data)

    def put_array(self, key, data, enc=None):
        e = enc or self.enc[key]
        if e[0] == "BYTE_ARRAY_STOP":
            assert bytes([e[1]]) not in data
            self.ext.setdefault(e[2], bytearray()).extend(data + bytes([e[1]]))
        else:
            self.put_int(None, len(data), e[1]); assert e[2][0] == "EXTERNAL"
            self.ext.setdefault(e[2][1], bytearray()).extend(data)

    def core(self):
        out = bytearray((len(self.bits) + 7) // 8)
        for i, b in enumerate(self.bits):
            if b: out[i >> 3] |= 0x80 >> (i & 7)
        return bytes(out)


def enc_bytes(e):
    kind = e[0]
    if kind == "NULL": return itf8(0) + itf8(0)
    if kind == "EXTERNAL": p = itf8(e[1]); return itf8(1) + itf8(len(p)) + p
    if kind == "HUFFMAN": p = array_itf8(e[1]) + array_itf8(e[2]); return itf8(3) + itf8(len(p)) + p
    if kind == "BYTE_ARRAY_LEN": p = enc_bytes(e[1]) + enc_bytes(e[2]); return itf8(4) + itf8(len(p)) + p
    if kind == "BYTE_ARRAY_STOP": p = bytes([e[1]]) + itf8(e[2]); return itf8(5) + itf8(len(p)) + p
    if kind == "BETA": p = itf8(e[1]) + itf8(e[2]); return itf8(6) + itf8(len(p)) + p
    if kind == "SUBEXP": p = itf8(e[1]) + itf8(e[2]); return itf8(7) + itf8(len(p)) + p
    if kind == "GAMMA": p = itf8(e[1]); return itf8(9) + itf8(len(p)) + p
    raise ValueError(kind)


def huffman_lengths(values):
    """code lengths of a small alphabet (a plain Huffman construction)"""
    from heapq import heapify, heappop, heappush
    freq = {}
    for v in values: freq[v] = freq.get(v, 0) + 1
    if len(freq) == 1: return sorted(freq), [0]
    heap = [(f, i, (s,)) for i, (s, f) in enumerate(sorted(freq.items()))]; heapify(heap); depth = {s: 0 for s in freq}; k = len(heap)
    while len(heap) > 1:
        a = heappop(heap); b = heappop(heap)
        for s in a[2] + b[2]: depth[s] += 1
        heappush(heap, (a[0] + b[0], k, a[2] + b[2])); k += 1
    syms = sorted(freq)
    return syms, [depth[s] for s in syms]


SHARED_BLOCK = False   # test switch: RN, IN and the first tag's values share one external block (the skip bookkeeping of the product must keep all of them: tests/test_cpu_cram.py)


def encode(bam_path, out_path, genome=None, slice_records=2500, rr=True, multi_ref=False, chains=True, embed_ref=False, variety=True, methods=None, qual_features=False, slices_per_container=1, version=(3, 0), name_method=None):
    """genome: {contig name: bytes, upper case}. rr = False writes every base into the file ('b' features: no genome needed to read it). multi_ref packs several
    references into one slice (RI series, absolute positions). embed_ref stores the slice's reference stretch in the file. variety = False: raw EXTERNAL only."""
    global NAME_METHOD
    NAME_METHOD = name_method
    text, refs, recs = read_bam(bam_path)
    rgs = CD.read_groups(text)
    out = bytearray(b"CRAM" + bytes(version) + b"oracle/cram_encode\0\0"[:20].ljust(20, b"\0"))
    hdr = struct.pack("<i", len(text)) + text.encode()
    out += container(0, 0, 0, 0, 0, 0, [block(0, 0, 0, hdr)], [0])
    # ---- slices: runs of one reference (or anything, for multi-reference slices) ----
    groups = []; cur = []
    for r in recs:
        key = r["ref_id"]
        if cur and ((not multi_ref and key != cur[0]["ref_id"]) or len(cur) >= slice_records): groups.append(cur); cur = []
        cur.append(r)
    if cur: groups.append(cur)
    counter = 0; crai = []
    for k in range(0, len(groups), max(1, slices_per_container)):
        gs = groups[k:k + max(1, slices_per_container)]; at = len(out)
        c, line = encode_container(gs, refs, rgs, genome, rr, multi_ref, chains, embed_ref, variety, counter, methods, qual_features); counter += sum(len(g) for g in gs)
        out += c; crai.append("%d\t%d\t%d\t%d\t%d\t%d\n" % (line[0], line[1], line[2], at, line[3], line[4]))   # (the container's first slice)
    out += eof_container()
    open(out_path, "wb").write(bytes(out))
    # the index `samtools index` writes for a CRAM (.crai: gzip text - reference, start, span, container offset, slice offset in the container, slice size)
    import gzip
    with gzip.open(out_path + ".crai", "wb") as f: f.write("".join(crai).encode())


def encode_slice(g, refs, rgs, genome, rr, multi_ref, chains, embed_ref, variety, counter, block_methods=None, qual_features=False):
    return encode_container([g], refs, rgs, genome, rr, multi_ref, chains, embed_ref, variety, counter, block_methods, qual_features)


def encode_container(gs, refs, rgs, genome, rr, multi_ref, chains, embed_ref, variety, counter, block_methods=None, qual_features=False):
    """one container of len(gs) slices: ONE compression header (preservation map, encodings, tag dictionary) for all of them, a landmark per slice"""
    # ---- per slice: reference, span, read groups, mate chains, CRAM flags ----
    P = []
    for g in gs:
        ref_ids = sorted({r["ref_id"] for r in g})
        slice_ref = ref_ids[0] if len(ref_ids) == 1 and not multi_ref else -2
        if slice_ref >= 0:
            start = min(r["pos"] for r in g); span = max(ref_end(r) for r in g) - start + 1
        else: start = span = 0
        rg_of = []
        for r in g:
            t = r["tags"]; rg = -1
            if t and t[-1][0] == b"RG" and t[-1][1] == ord("Z") and t[-1][2][:-1].decode() in rgs: rg = rgs.index(t[-1][2][:-1].decode())
            rg_of.append(rg)
        n = len(g); link = [None] * n; cf = [0] * n
        if chains:   # chains where the decoder's rules give back the BAM's fields
            open_by_name = {}
            for i, r in enumerate(g):
                j = open_by_name.pop(r["name"], None)
                if j is None:
                    if r["flag"] & 1: open_by_name[r["name"]] = i
                    continue
                if chain_reproduces(g[j], g[i]): link[j] = i; cf[j] |= CD.CF_MATE_DOWNSTREAM; cf[i] |= 0x100   # (0x100: marks the last member; not written)
        for i, r in enumerate(g):
            if cf[i] & (CD.CF_MATE_DOWNSTREAM | 0x100): continue
            plain = not r["flag"] & 1 and r["mate_ref"] == -1 and r["mate_pos"] == 0 and r["tlen"] == 0 and not r["flag"] & 0x28
            if not plain: cf[i] |= CD.CF_DETACHED
        for i, r in enumerate(g):
            cf[i] &= 0xff
            if len(r["seq"]) == 0: cf[i] |= CD.CF_NO_SEQ
            elif qual_features and i % 5 == 2 and not r["flag"] & 4 and len(r["seq"]) > 30: pass     # (a lossy-quality record: single qualities as features below, no array)
            elif r["qual"] != b"\xff" * len(r["qual"]): cf[i] |= CD.CF_QUAL_ARRAY
        P.append(dict(g=g, ref=slice_ref, start=start, span=span, rg_of=rg_of, link=link, cf=cf))
    ap_delta = all(p["ref"] != -2 for p in P)      # (the AP flag of the preservation map holds for the whole container)
    # ---- series encodings of the container ----
    ids = {}
    def ext(key):
        ids.setdefault(key, len(ids) + 1); return ("EXTERNAL", ids[key])
    E = {k: ext(k) for k in ("BF", "RL", "AP", "NP", "TS", "NF", "TL", "FP", "BS", "BA", "QS", "RI", "MF", "NS", "HC", "PD", "RS", "FC")}
    E["RN"] = ("BYTE_ARRAY_STOP", 0, ext("RN")[1]); E["IN"] = ("BYTE_ARRAY_STOP", 0, ext("IN")[1]); E["SC"] = ("BYTE_ARRAY_STOP", 0, ext("SC")[1])
    if SHARED_BLOCK: E["RN"] = ("BYTE_ARRAY_STOP", 0, ext("IN")[1])   # (a legal layout htslib does not write: read names, inserted bases and - below - one tag's values in ONE external block)
    E["BB"] = ("BYTE_ARRAY_LEN", ext("BBl"), ext("BBv")); E["QQ"] = ("BYTE_ARRAY_LEN", ext("QQl"), ext("QQv"))
    all_rg = [x for p in P for x in p["rg_of"]]; all_cf = [x for p in P for x in p["cf"]]
    if variety:
        E["RG"] = ("HUFFMAN",) + tuple(huffman_lengths(all_rg))
        E["MQ"] = ("BETA", 0, 8); E["FN"] = ("GAMMA", 1); E["DL"] = ("SUBEXP", 0, 2)
        E["CF"] = ("HUFFMAN",) + tuple(huffman_lengths(all_cf))
    else:
        E["RG"] = ext("RG"); E["MQ"] = ext("MQ"); E["FN"] = ext("FN"); E["DL"] = ext("DL"); E["CF"] = ext("CF")
    # ---- tags ----
    TD = []; tag_enc = {}
    for p in P:
        p["tl_of"] = []
        for r, rg in zip(p["g"], p["rg_of"]):
            tags = r["tags"][:-1] if rg >= 0 else r["tags"]
            line = tuple((t, typ) for t, typ, _ in tags)
            if line not in TD: TD.append(line)
            p["tl_of"].append(TD.index(line))
            for t, typ, _ in tags:
                key = (t[0] << 16) | (t[1] << 8) | typ
                if key not in tag_enc: tag_enc[key] = ("BYTE_ARRAY_LEN", ext("tl%d" % key), ext("IN") if SHARED_BLOCK and not tag_enc else ext("tv%d" % key))
    sm = SUBST_DEFAULT
    subst_code = {}
    for ri, rb in enumerate(BASES):
        others = [b for b in BASES if b != rb]
        for k, b in enumerate(others): subst_code[(rb, b)] = (sm[ri] >> (6 - 2 * k)) & 3

    def ref_base(r, p0):
        seq = genome[refs[r["ref_id"]][0]]
        return chr(seq[p0]) if 0 <= p0 < len(seq) else "N"
    # ---- compression header ----
    pres = b"RN\x01" + b"AP" + bytes([1 if ap_delta else 0]) + b"RR" + bytes([1 if rr else 0]) + b"SM" + sm
    td = b"".join(b"".join(t + bytes([typ]) for t, typ in line) + b"\0" for line in TD)
    pres += b"TD" + itf8(len(td)) + td
    pres = itf8(5) + pres
    dsm = itf8(len(E)) + b"".join(k.encode() + enc_bytes(e) for k, e in E.items())
    tgm = itf8(len(tag_enc)) + b"".join(itf8(k) + enc_bytes(e) for k, e in tag_enc.items())
    ch = itf8(len(pres)) + pres + itf8(len(dsm)) + dsm + itf8(len(tgm)) + tgm
    ch_block = block(0, 1, 0, ch)
    blocks = [ch_block]; landmarks = []; total_bases = 0; at = len(ch_block); first_slice_size = 0; rec_counter = counter
    # ---- slices ----
    for p in P:
        g, slice_ref, start, span, rg_of, link, cf, tl_of = p["g"], p["ref"], p["start"], p["span"], p["rg_of"], p["link"], p["cf"], p["tl_of"]
        W = SliceWriter(E); prev = start; bases = 0
        embedded = None
        if embed_ref and slice_ref >= 0 and rr: embedded = genome[refs[slice_ref][0]][start - 1:start - 1 + span]
        for i, r in enumerate(g):
            mapped_rec = not r["flag"] & 4
            rl = len(r["seq"]) if len(r["seq"]) else sum(k for op, k in r["cigar"] if op in "MIS=X")
            W.put_int("BF", r["flag"]); W.put_int("CF", cf[i])
            if slice_ref == -2: W.put_int("RI", r["ref_id"])
            W.put_int("RL", rl)
            if ap_delta: W.put_int("AP", r["pos"] - prev); prev = r["pos"]
            else: W.put_int("AP", r["pos"])
            W.put_int("RG", rg_of[i]); W.put_array("RN", r["name"])
            if cf[i] & CD.CF_DETACHED:
                W.put_int("MF", (1 if r["flag"] & 0x20 else 0) | (2 if r["flag"] & 0x8 else 0))
                W.put_int("NS", r["mate_ref"]); W.put_int("NP", r["mate_pos"]); W.put_int("TS", r["tlen"])
            elif cf[i] & CD.CF_MATE_DOWNSTREAM: W.put_int("NF", link[i] - i - 1)
            W.put_int("TL", tl_of[i])
            for t, typ, v in (r["tags"][:-1] if rg_of[i] >= 0 else r["tags"]):
                W.put_array(None, v, tag_enc[(t[0] << 16) | (t[1] << 8) | typ])
            bases += rl
            if mapped_rec:
                feats = []; rp = 0; gp = r["pos"] - 1; seq = r["seq"]
                for op, k in r["cigar"]:
                    if op in "M=X":
                        if not rr:
                            if len(seq): feats.append(("b", rp + 1, seq[rp:rp + k]))
                        elif len(seq):
                            for x in range(k):
                                b = chr(seq[rp + x]); rb = ref_base(r, gp + x)
                                if rb not in BASES: rb = "N"
                                if b == rb: continue
                                if b in BASES: feats.append(("X", rp + x + 1, subst_code[(rb, b)]))
                                else: feats.append(("B", rp + x + 1, (seq[rp + x], r["qual"][rp + x])))
                        rp += k; gp += k
                    elif op == "I":
                        feats.append(("I", rp + 1, seq[rp:rp + k]) if k > 1 or not variety else ("i", rp + 1, seq[rp])); rp += k
                    elif op == "S": feats.append(("S", rp + 1, seq[rp:rp + k])); rp += k
                    elif op == "D": feats.append(("D", rp + 1, k)); gp += k
                    elif op == "N": feats.append(("N", rp + 1, k)); gp += k
                    elif op == "H": feats.append(("H", rp + 1, k))
                    elif op == "P": feats.append(("P", rp + 1, k))
                if qual_features and not cf[i] & CD.CF_QUAL_ARRAY and not cf[i] & CD.CF_NO_SEQ and len(seq) > 30:
                    feats += [("Q", 3, r["qual"][2]), ("q", 10, r["qual"][9:17]), ("Q", len(seq), r["qual"][-1])]
                    feats.sort(key=lambda f: (f[1], 0 if f[0] in "Qq" else 1))
                W.put_int("FN", len(feats)); last = 0
                for code, fp, v in feats:
                    W.put_byte("FC", ord(code)); W.put_int("FP", fp - last); last = fp
                    if code == "B": W.put_byte("BA", v[0]); W.put_byte("QS", v[1])
                    elif code == "X": W.put_byte("BS", v)
                    elif code == "I": W.put_array("IN", v)
                    elif code == "S": W.put_array("SC", v)
                    elif code == "i": W.put_byte("BA", v)
                    elif code == "b": W.put_array("BB", v)
                    elif code == "Q": W.put_byte("QS", v)
                    elif code == "q": W.put_array("QQ", v)
                    elif code == "D": W.put_int("DL", v)
                    elif code == "N": W.put_int("RS", v)
                    elif code == "H": W.put_int("HC", v)
                    elif code == "P": W.put_int("PD", v)
                W.put_int("MQ", r["mapq"])
                if cf[i] & CD.CF_QUAL_ARRAY: W.put_bytes("QS", r["qual"])
            else:
                if not cf[i] & CD.CF_NO_SEQ: W.put_bytes("BA", r["seq"])
                if cf[i] & CD.CF_QUAL_ARRAY: W.put_bytes("QS", r["qual"])
        # ---- the slice's blocks ----
        ext_blocks = []; content_ids = []
        methods = block_methods or ([0, 1, 4, 41] if variety else [0])
        for k, (cid, data) in enumerate(sorted(W.ext.items())):
            m = methods[k % len(methods)]
            if cid == ids.get("QS") and variety and not block_methods: m = 41
            if m in (4, 41) and len(data) > 400000: m = 1
            if NAME_METHOD is not None and cid == ids.get("RN"): m = NAME_METHOD
            ext_blocks.append(block(m, 4, cid, bytes(data))); content_ids.append(cid)
        emb_id = -1
        if embedded is not None:
            emb_id = max(content_ids + [0]) + 1; ext_blocks.append(block(1, 4, emb_id, bytes(embedded))); content_ids.append(emb_id)
        core = block(0, 5, 0, W.core())
        md5 = b"\0" * 16
        if slice_ref >= 0 and rr:
            seq = genome[refs[slice_ref][0]]; md5 = hashlib.md5(bytes(seq[start - 1:start - 1 + span])).digest()
        sh = itf8(slice_ref) + itf8(start) + itf8(span) + itf8(len(g)) + ltf8(rec_counter) + itf8(1 + len(ext_blocks)) + array_itf8(content_ids) + itf8(emb_id) + md5
        sblocks = [block(0, 2, 0, sh), core] + ext_blocks
        landmarks.append(at); size = sum(len(b) for b in sblocks); at += size
        if not first_slice_size: first_slice_size = size
        blocks += sblocks; total_bases += bases; rec_counter += len(g)
    crefs = {p["ref"] for p in P}
    if len(crefs) == 1 and -2 not in crefs and next(iter(crefs)) >= 0:
        cref = next(iter(crefs)); cstart = min(p["start"] for p in P); cspan = max(p["start"] + p["span"] for p in P) - cstart
    elif crefs == {-1}: cref, cstart, cspan = -1, 0, 0
    else: cref, cstart, cspan = -2, 0, 0
    n_rec = sum(len(p["g"]) for p in P)
    return container(cref, cstart, cspan, n_rec, counter, total_bases, blocks, landmarks), (P[0]["ref"], P[0]["start"], P[0]["span"], len(ch_block), first_slice_size)


def chain_reproduces(a, b):
    """two records of one template, a in front of b in the slice: do the decoder's chain rules (cram_decode.resolve_mates) give back their mate fields?"""
    for x, y in ((a, b), (b, a)):
        if x["mate_ref"] != y["ref_id"] or x["mate_pos"] != y["pos"]: return False
        if bool(x["flag"] & 0x20) != bool(y["flag"] & 0x10) or bool(x["flag"] & 0x8) != bool(y["flag"] & 0x4) or not x["flag"] & 1: return False
    if a["ref_id"] != b["ref_id"]: want = (0, 0)
    else:
        left = min(a["pos"], b["pos"]); right = max(ref_end(a), ref_end(b)); tlen = right - left + 1
        cnt = (a["pos"] == left) + (b["pos"] == left)
        sign = lambda r: tlen if r["pos"] == left and (cnt == 1 or r["flag"] & 0x40) else -tlen
        want = (sign(a), sign(b))
    if a["flag"] & 4 or b["flag"] & 4: want = (0, 0)
    return (a["tlen"], b["tlen"]) == want
