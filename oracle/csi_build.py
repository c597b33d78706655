"""ORACLE (test infrastructure, never imported by the product): CPU restatement of CSI index construction and of the CSI region query.

CSI (hts-specs CSIv1: "Coordinate Sorted Index") is the BAI binning scheme with two free parameters - min_shift (width of the smallest bin, 14 in
BAI) and depth (number of levels below bin 0, 5 in BAI) - and, per bin, the field loff in place of BAI's per-reference linear index:
    magic "CSI\\1", int32 min_shift, int32 depth, int32 l_aux, aux[l_aux], int32 n_ref,
    per reference: int32 n_bin, per bin: uint32 bin, uint64 loff, int32 n_chunk, n_chunk x (uint64 beg, uint64 end)
    optional uint64 n_no_coor;   the whole file inside a BGZF container.
htslib builds it with the same hts_idx_push / hts_idx_finish / compress_binning as a BAI (hts.c; restated in oracle/bai_build.py, which is pinned on
the reference's htslib-written .bai fixtures); what differs is restated here:
  * sam_index_build3(fn, fnidx, min_shift > 0): depth = the smallest n with (longest reference + 256) <= 2^(min_shift + 3 n)   (sam.c sam_index)
  * update_loff: loff(bin) = linear index at the bin's first window, 0 for the pseudo-bin and for bins behind the last window (bai_build.update_loff)
  * hts_idx_save: loff per bin, no linear index, BGZF-compressed
  * hts_itr_query: the lower bound of a query comes from loff of the bottom-level bin of the region's start, or of the nearest bin in front of it /
    above it that exists (csi_min_off)
PARITY PINNING: the reference holds no .csi file (its fixtures are .bai and .crai), so this file cannot be compared with an htslib-written CSI. It is
pinned one step removed: for geometry (14, 5) the bins and chunks built here must equal every htslib-written .bai fixture's, and loff must equal that
fixture's linear index at the bin's first window (tests/test_oracle_bai.py::test_csi_at_bai_geometry_equals_the_fixture_indices); other geometries
are checked against a sequential pass over the BAM (every overlapping record inside the queried range).
"""
import struct
import zlib

import bai_build as B


def ref_lengths(path):
    img = open(path, "rb").read()
    pos = 0; stream = bytearray()
    while pos < len(img):
        bs = struct.unpack_from("<H", img, pos + 16)[0] + 1
        stream += zlib.decompress(img[pos + 18:pos + bs - 8], -15); pos += bs
        if len(stream) >= 12:
            l_text = struct.unpack_from("<i", stream, 4)[0]
            if len(stream) >= 12 + l_text:
                n_ref = struct.unpack_from("<i", stream, 8 + l_text)[0]
                o = 12 + l_text; lens = []; ok = True
                for _ in range(n_ref):
                    if o + 4 > len(stream): ok = False; break
                    l_name = struct.unpack_from("<i", stream, o)[0]
                    if o + 8 + l_name > len(stream): ok = False; break
                    lens.append(struct.unpack_from("<i", stream, o + 4 + l_name)[0]); o += 8 + l_name
                if ok: return lens
    raise ValueError("truncated BAM header")


def depth_for(ref_lens, min_shift):
    """sam.c sam_index: for (n_lvls = 0, s = 1 << min_shift; max_len > s; ++n_lvls, s <<= 3)"""
    max_len = max(list(ref_lens) + [0]) + 256
    n, s = 0, 1 << min_shift
    while max_len > s: n += 1; s <<= 3
    return n


def build_for_bam(path, min_shift=14, depth=None):
    """-> (geom, [per reference {bin: (loff, [[beg, end], ...])}], n_no_coor)"""
    n_ref, offset0, recs, final = B.read_bam(path)
    if depth is None: depth = depth_for(ref_lengths(path), min_shift)
    geom = (min_shift, depth)
    return from_index(B.build(n_ref, offset0, recs, final["eof_block"], "backward", geom), geom)


def from_index(ix, geom):
    refs = []
    for t, b in enumerate(ix.bidx):
        refs.append({} if b is None else {k: (ix.loff[t][k], v) for k, v in b.items()})
    return geom, refs, ix.n_no_coor


def serialize(csi):
    (min_shift, depth), refs, n_no_coor = csi
    out = bytearray(b"CSI\1" + struct.pack("<iiii", min_shift, depth, 0, len(refs)))
    for bins in refs:
        out += struct.pack("<i", len(bins))
        for b in sorted(bins):
            loff, chunks = bins[b]
            out += struct.pack("<IQi", b, loff, len(chunks))
            for beg, end in chunks: out += struct.pack("<QQ", beg, end)
    if n_no_coor is not None: out += struct.pack("<Q", n_no_coor)
    return bytes(out)


def bgzf(data, level=6):
    out = bytearray()
    for o in list(range(0, len(data), 0xff00)) + [None]:
        piece = data[o:o + 0xff00] if o is not None else b""
        c = zlib.compressobj(level, zlib.DEFLATED, -15); z = c.compress(piece) + c.flush()
        out += struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(z) + 25) + z + struct.pack("<II", zlib.crc32(piece), len(piece))
    return bytes(out)


def unbgzf(d):
    if d[:2] != b"\x1f\x8b": return d
    out = bytearray(); pos = 0
    while pos < len(d):
        bs = struct.unpack_from("<H", d, pos + 16)[0] + 1
        out += zlib.decompress(d[pos + 18:pos + bs - 8], -15); pos += bs
    return bytes(out)


def write_csi(path, csi, compress=True):
    raw = serialize(csi)
    open(path, "wb").write(bgzf(raw) if compress else raw)


def parse_csi(path):
    d = unbgzf(open(path, "rb").read())
    assert d[:4] == b"CSI\1"
    min_shift, depth, l_aux = struct.unpack_from("<iii", d, 4); o = 16 + l_aux
    n_ref = struct.unpack_from("<i", d, o)[0]; o += 4; refs = []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", d, o)[0]; o += 4; bins = {}
        for _ in range(n_bin):
            b, loff, n_chunk = struct.unpack_from("<IQi", d, o); o += 16
            bins[b] = (loff, [list(struct.unpack_from("<QQ", d, o + 16 * i)) for i in range(n_chunk)]); o += 16 * n_chunk
        refs.append(bins)
    return (min_shift, depth), refs, (struct.unpack_from("<Q", d, o)[0] if o + 8 <= len(d) else None)


def reg2bins(beg, end, geom):
    """hts.c reg2bins: every bin that can hold a record overlapping [beg, end) (0-based, half open)"""
    min_shift, depth = geom
    if beg >= end: return []
    end = min(end, B.max_pos(geom)) - 1
    bins = []; s = min_shift + 3 * depth; t = 0
    for l in range(depth + 1):
        bins.extend(range(t + (beg >> s), t + (end >> s) + 1))
        s -= 3; t += 1 << (3 * l)
    return bins


def csi_min_off(bins, beg, geom):
    """hts_itr_query: loff of the bottom-level bin of beg; when that bin does not exist, of the sibling in front of it, else of the parent, and so on"""
    b = B.bin_first(geom[1]) + (beg >> geom[0])
    while b:
        if b in bins: break
        first = (((b - 1) >> 3) << 3) + 1
        b = b - 1 if b > first else (b - 1) >> 3
    return bins[b][0] if b in bins else 0


def query(csi, tid, beg, end):
    """-> sorted chunks [(beg, end)] whose records may overlap [beg, end) of reference tid (what the iterator of hts_itr_query reads), cut at the lower bound"""
    geom, refs, _ = csi
    if tid < 0 or tid >= len(refs) or beg >= B.max_pos(geom): return []
    bins = refs[tid]; lo = csi_min_off(bins, beg, geom); out = []
    for b in reg2bins(beg, end, geom):
        if b in bins:
            out.extend((max(c[0], lo), c[1]) for c in bins[b][1] if c[1] > lo)
    return sorted(out)
