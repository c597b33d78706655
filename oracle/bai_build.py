"""ORACLE (test infrastructure, never imported by the product): CPU restatement of BAI index construction.

The reference never writes an index itself - its tests and tools consume the .bai files that `samtools index` (htslib's
sam_index_build -> hts_idx_push / hts_idx_finish / compress_binning / update_loff, hts.c) produced. htslib is a third-party
dependency that is absent from /root/reference (Makefile:259-267 unpacks htslib/htslib_linux.zip), so this file restates the
published algorithm (SAM spec section 5.2 + hts.c) and is PINNED on the reference's own fixtures: for every BAM + .bai pair
under tests/golden/ref_in that an htslib wrote, the index built here equals the fixture index (same bins, chunks, linear
index, pseudo-bin and n_no_coor; tests/test_oracle_bai.py). The fixtures span three htslib generations that differ in two
details, both restated here as variants:
  fill   how linear-index windows without a read are filled: "backward" (current htslib: from the next window that has
         one) or "forward" (older: from the previous one; leading windows take the start of the reference's records)
  final  where the last chunk ends: "eof_block" (current: the address of the EOF block) or "file_end" (oldest: behind it)
CURRENT = ("backward", "eof_block") is what the product writes. Three fixture indices were written by pre-htslib samtools
0.1.x (no n_no_coor field, different chunk starts); they are only used for query tests.
Record by record, pure Python: fixture-sized inputs only.
"""
import struct
import zlib

MIN_SHIFT, N_LVLS = 14, 5
BAI_GEOM = (MIN_SHIFT, N_LVLS)                   # (min_shift, n_lvls); a CSI index carries its own pair (oracle/csi_build.py)
N_BINS = ((1 << (3 * N_LVLS + 3)) - 1) // 7      # 37449
META_BIN = N_BINS + 1                            # 37450
MAX_POS = 1 << (MIN_SHIFT + 3 * N_LVLS)             # 2^29
MIN_MARKER_DIST = 0x10000
UNSET = (1 << 64) - 1


def n_bins(geom):
    return ((1 << (3 * geom[1] + 3)) - 1) // 7


def meta_bin(geom):
    return n_bins(geom) + 1


def max_pos(geom):
    return 1 << (geom[0] + 3 * geom[1])


def reg2bin(beg, end, geom=BAI_GEOM):
    """hts_reg2bin (SAM spec 5.3; min_shift 14 / 5 levels for BAI); Python's >> is arithmetic like C's on the signed values htslib passes"""
    end -= 1
    s, t = geom[0], ((1 << (3 * geom[1])) - 1) // 7
    l = geom[1]
    while l > 0:
        if beg >> s == end >> s:
            return t + (beg >> s)
        l -= 1; s += 3; t -= 1 << (3 * l)
    return 0


def bin_first(l):
    return ((1 << (3 * l)) - 1) // 7


def read_bam(path):
    """-> (n_ref, offset0, [(tid, pos, endpos, voff_after, mapped)], final voff): what sam_index_build sees - bgzf_tell after the header, and after every
    record (a position at the end of a block is reported as offset 0 of the block that follows, bgzf.c bgzf_read)"""
    img = open(path, "rb").read()
    pos = 0; members = []; stream = bytearray()
    while pos < len(img):
        bs = struct.unpack_from("<H", img, pos + 16)[0] + 1
        raw = zlib.decompress(img[pos + 18:pos + bs - 8], -15)
        members.append((pos, len(stream), len(raw), pos + bs)); stream += raw; pos += bs
    data = [m for m in members if m[2]]

    def tell(u):
        lo, hi = 0, len(data) - 1
        while lo < hi:
            mid = (lo + hi + 1) // 2
            if data[mid][1] < u: lo = mid      # the member that holds byte u - 1
            else: hi = mid - 1
        off, up, n, nxt = data[lo]
        return (nxt << 16) if u - up == n else (off << 16) | (u - up)

    assert stream[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", stream, 4)[0]; o = 8 + l_text
    n_ref = struct.unpack_from("<i", stream, o)[0]; o += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", stream, o)[0]; o += 8 + l_name
    offset0 = tell(o); recs = []
    while o < len(stream):
        bs, tid, p, l_name, mapq, b, n_cig, flag, l_seq = struct.unpack_from("<iiiBBHHHi", stream, o)
        if tid >= n_ref or tid < -1: raise ValueError("reference id outside the header (sam_read1 fails with ERANGE)")
        rlen = 0
        if not flag & 4:
            co = o + 36 + l_name
            for k in range(n_cig):
                c = struct.unpack_from("<I", stream, co + 4 * k)[0]
                if (c & 15) in (0, 2, 3, 7, 8): rlen += c >> 4
        o += 4 + bs
        recs.append((tid, p, p + (rlen if rlen else 1), tell(o), not flag & 4))
    # the failed read at the end finds the EOF block empty: the reader stays at that block's address (oldest htslib: behind it)
    return n_ref, offset0, recs, {"eof_block": tell(len(stream)), "file_end": members[-1][3] << 16}


class Index:
    def __init__(self, n_ref):
        self.bidx = [None] * n_ref; self.lidx = [[] for _ in range(n_ref)]; self.n_no_coor = 0


CURRENT = ("backward", "eof_block")
VARIANTS = [CURRENT, ("forward", "eof_block"), ("forward", "file_end")]


def build(n_ref, offset0, recs, final, fill="backward", geom=BAI_GEOM):
    """hts_idx_push for every record, then hts_idx_finish (hts.c)"""
    ix = Index(n_ref)
    MIN_SHIFT = geom[0]; MAX_POS = max_pos(geom); META_BIN = meta_bin(geom)
    last_bin = save_bin = 0xffffffff
    last_off = save_off = off_beg = off_end = offset0
    n_mapped = n_unmapped = 0; last_coor = 0xffffffff
    last_tid = 0xffffffff; save_tid = 0xffffffff

    def insert_b(tid, b, beg, end):
        ix.bidx[tid].setdefault(b, []).append([beg, end])

    for tid, beg, end, offset, mapped in recs:
        if tid < 0: beg, end = -1, 0
        if last_tid != tid or (last_tid != 0xffffffff and last_tid >= 0 and tid < 0):
            if tid >= 0 and ix.n_no_coor: raise ValueError("NO_COOR reads not in a single block at the end")
            if tid >= 0 and ix.bidx[tid] is not None: raise ValueError("chromosome blocks not continuous")
            last_tid = tid; last_bin = 0xffffffff
        elif tid >= 0 and last_coor > beg:
            raise ValueError("unsorted positions")
        if end < beg: end = beg + 1
        if tid >= 0:
            if beg > MAX_POS or end > MAX_POS: raise ValueError("region cannot be stored in the index (2^%d limit)" % (geom[0] + 3 * geom[1]))
            if ix.bidx[tid] is None: ix.bidx[tid] = {}
            if beg < 0: beg = 0
            if end <= 0: end = 1
            L = ix.lidx[tid]; b0 = beg >> MIN_SHIFT; e0 = (end - 1) >> MIN_SHIFT
            if len(L) < e0 + 1: L.extend([UNSET] * (e0 + 1 - len(L)))
            for i in range(b0, e0 + 1):
                if L[i] == UNSET: L[i] = last_off
        else:
            ix.n_no_coor += 1
        b = reg2bin(beg, end, geom)
        if last_bin != b:
            if save_bin != 0xffffffff: insert_b(save_tid, save_bin, save_off, last_off)
            if last_bin == 0xffffffff and save_bin != 0xffffffff:
                off_end = last_off
                insert_b(save_tid, META_BIN, off_beg, off_end)
                insert_b(save_tid, META_BIN, n_mapped, n_unmapped)
                n_mapped = n_unmapped = 0; off_beg = off_end
            save_off = last_off; save_bin = last_bin = b; save_tid = tid
        if mapped: n_mapped += 1
        else: n_unmapped += 1
        last_off = offset; last_coor = beg
    # hts_idx_finish
    if save_tid != 0xffffffff and save_tid >= 0:
        insert_b(save_tid, save_bin, save_off, final)
        insert_b(save_tid, META_BIN, off_beg, final)
        insert_b(save_tid, META_BIN, n_mapped, n_unmapped)
    for t in range(n_ref):
        update_loff(ix, t, fill, geom); compress_binning(ix, t, geom)
    return ix


def bin_bot(b, geom):
    """hts_bin_bot: the first window (bottom-level slot) of a bin"""
    l = 0; x = b
    while x: l += 1; x = (x - 1) >> 3
    return (b - bin_first(l)) << ((geom[1] - l) * 3)


def update_loff(ix, t, fill, geom=BAI_GEOM):
    """fills the windows without a read, then sets every bin's loff = the linear index at the bin's first window (0 for the pseudo-bin and for a bin
    behind the last window) - before compress_binning, as hts_idx_finish does; a CSI file stores loff instead of the linear index"""
    L = ix.lidx[t]; META_BIN = meta_bin(geom)
    if fill == "backward":
        # the last entry is always valid
        for l in range(len(L) - 2, -1, -1):
            if L[l] == UNSET: L[l] = L[l + 1]
    else:
        B = ix.bidx[t]; off0 = B[META_BIN][0][0] if B and META_BIN in B else 0
        l = 0
        while l < len(L) and L[l] == UNSET: L[l] = off0; l += 1
        for l in range(1, len(L)):
            if L[l] == UNSET: L[l] = L[l - 1]
    if ix.bidx[t] is not None:
        if not hasattr(ix, "loff"): ix.loff = [None] * len(ix.bidx)
        ix.loff[t] = {}
        for b in ix.bidx[t]:
            bot = bin_bot(b, geom) if b < n_bins(geom) else None
            ix.loff[t][b] = L[bot] if bot is not None and bot < len(L) else 0


def compress_binning(ix, t, geom=BAI_GEOM):
    B = ix.bidx[t]
    if B is None: return
    N_LVLS = geom[1]; N_BINS = n_bins(geom)
    for l in range(N_LVLS, 0, -1):
        start = bin_first(l)
        for k in sorted(B.keys()):
            if k >= N_BINS or k < start or k not in B: continue
            p = B[k]
            if l < N_LVLS and len(p) > 1: p.sort(key=lambda c: c[0])
            if (p[-1][1] >> 16) - (p[0][0] >> 16) < MIN_MARKER_DIST:
                kp = (k - 1) >> 3
                if kp not in B: continue
                B[kp].extend(p); del B[k]
    if 0 in B: B[0].sort(key=lambda c: c[0])
    for k, p in B.items():
        if k >= N_BINS: continue
        m = 0
        for l in range(1, len(p)):
            if p[m][1] >> 16 >= p[l][0] >> 16:
                if p[m][1] < p[l][1]: p[m][1] = p[l][1]
            else:
                m += 1; p[m] = p[l]
        del p[m + 1:]


def parse_bai(path):
    """-> (per reference ({bin: [[beg, end], ...]}, [ioffset]), n_no_coor or None)"""
    d = open(path, "rb").read()
    assert d[:4] == b"BAI\1"
    n_ref = struct.unpack_from("<i", d, 4)[0]; o = 8; refs = []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", d, o)[0]; o += 4; bins = {}
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", d, o); o += 8
            bins[b] = [list(struct.unpack_from("<QQ", d, o + 16 * i)) for i in range(n_chunk)]; o += 16 * n_chunk
        n_intv = struct.unpack_from("<i", d, o)[0]; o += 4
        refs.append((bins, list(struct.unpack_from("<%dQ" % n_intv, d, o)))); o += 8 * n_intv
    return refs, (struct.unpack_from("<Q", d, o)[0] if o + 8 <= len(d) else None)


def as_parsed(ix):
    return [((b if b is not None else {}), l) for b, l in zip(ix.bidx, ix.lidx)], ix.n_no_coor


def build_for_bam(path, variant=CURRENT):
    n_ref, offset0, recs, final = read_bam(path)
    return as_parsed(build(n_ref, offset0, recs, final[variant[1]], variant[0]))


def device_view(n_ref, offset0, recs, cuts=(), geom=BAI_GEOM):
    """What the device half of ngsqc_write_bai hands to the host half (ngsqc_bai_assemble), computed from the record list: runs (a record whose
    (reference, bin) differs from its predecessor's; the first record behind every cut = tile boundary starts one too, and the record in front of a cut is
    reported as a kind-1 entry), first start offset per 16 kb window, mapped / unmapped counts. -> (runs, lidx, lidx_first, counts)"""
    MIN_SHIFT = geom[0]
    nwin = [0] * n_ref
    for tid, beg, end, _, _ in recs:
        if tid >= 0: nwin[tid] = max(nwin[tid], ((max(end, 1) - 1) >> MIN_SHIFT) + 1)
    first = [0]
    for t in range(n_ref): first.append(first[-1] + nwin[t] + 3)
    lidx = [UNSET] * first[-1]; counts = [0] * (2 * (n_ref + 1)); runs = []
    start = offset0; prev = None; cuts = set(cuts)
    for i, (tid, beg, end, after, mapped) in enumerate(recs):
        if tid < 0: tid, beg, end = -1, -1, 0
        else:
            beg = max(beg, 0); end = max(end, 1)
            for w in range(beg >> MIN_SHIFT, ((end - 1) >> MIN_SHIFT) + 1):
                lidx[first[tid] + w] = min(lidx[first[tid] + w], start)
        key = (tid, reg2bin(beg, end, geom))
        if key != prev or i in cuts: runs.append((start, tid, key[1], recs[i][1], 0))
        if i + 1 in cuts or i + 1 == len(recs): runs.append((start, tid, key[1], max(recs[i][1], 0), 1))
        counts[2 * (tid if tid >= 0 else n_ref) + (0 if mapped else 1)] += 1
        prev = key; start = after
    return runs, lidx, first, counts


def write_bai(path, parsed):
    """serialize (refs, n_no_coor) as parse_bai returns it (bins in ascending order)"""
    refs, n_no_coor = parsed
    out = bytearray(b"BAI\1" + struct.pack("<i", len(refs)))
    for bins, lidx in refs:
        out += struct.pack("<i", len(bins))
        for b in sorted(bins):
            out += struct.pack("<Ii", b, len(bins[b]))
            for beg, end in bins[b]:
                out += struct.pack("<QQ", beg, end)
        out += struct.pack("<i", len(lidx)) + struct.pack("<%dQ" % len(lidx), *lidx)
    if n_no_coor is not None:
        out += struct.pack("<Q", n_no_coor)
    open(path, "wb").write(bytes(out))
