"""K1 (BGZF inflate on the GPU) against zlib on every DEFLATE shape a BAM writer can emit: stored members (samtools -u /
level 0), fixed-Huffman blocks, dynamic blocks at levels 1/6/9, Z_RLE (overlapping matches, distance 1), Z_HUFFMAN_ONLY
(literal-only), tiny and ragged members, several deflate blocks per member, empty members in the middle of the file,
and corrupted payloads (must fail with an error, never hang or crash). Bit-exact on the inflated stream; the K1 switches
(no-parking phase 1, tiny tiles, CRC check off) run on the same inputs. The CRC32 of every member is verified on the GPU
(htslib does: a payload that is damaged but still inflates to the right length must raise the reference's read error)."""
import gzip
import struct
import zlib

import numpy as np
import pytest

import bamgen_lib as G

pytestmark = pytest.mark.gpu
ngsqc = __import__("importlib").import_module("ngs-bits_amd")

EOF_MEMBER = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def member(payload_raw: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0) -> bytes:
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    if flush_every:
        comp = b"".join(co.compress(payload_raw[i:i + flush_every]) + co.flush(zlib.Z_FULL_FLUSH) for i in range(0, len(payload_raw), flush_every))
        comp += co.flush()
    else:
        comp = co.compress(payload_raw) + co.flush()
    bsize = 18 + len(comp) + 8
    assert bsize <= 65536, bsize
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize - 1) + comp
            + struct.pack("<II", zlib.crc32(payload_raw) & 0xFFFFFFFF, len(payload_raw)))


def rebgzf(raw: bytes, sizes, **kw) -> bytes:
    out, pos, k = [], 0, 0
    while pos < len(raw):
        n = sizes[k % len(sizes)]; k += 1
        out.append(member(raw[pos:pos + n], **kw)); pos += n
    return b"".join(out) + EOF_MEMBER


@pytest.fixture(scope="module")
def raw_bam():
    img = G.generate(30000, seed=5, threads=2).tobytes()
    return gzip.decompress(img)


def _roundtrip(raw, image):
    h = ngsqc.Handle(data=np.frombuffer(image, dtype=np.uint8))
    try:
        h.decode()
        got = h.inflated()
        assert got.size == len(raw)
        assert np.array_equal(got, np.frombuffer(raw, dtype=np.uint8))
        return h.n_records
    finally:
        h.close()


SHAPES = {
    "stored": dict(level=0, sizes=[60000]),
    "stored_small": dict(level=0, sizes=[1, 7, 300, 65000, 2]),
    "fixed": dict(level=6, strategy=zlib.Z_FIXED, sizes=[40000]),
    "level1": dict(level=1, sizes=[65280]),
    "level6_ragged": dict(level=6, sizes=[65280, 1, 13, 4097, 30000, 64]),
    "level9": dict(level=9, sizes=[65280]),
    "rle": dict(level=6, strategy=zlib.Z_RLE, sizes=[50000]),
    "huffman_only": dict(level=6, strategy=zlib.Z_HUFFMAN_ONLY, sizes=[30000]),
    "multi_block": dict(level=6, sizes=[65280], flush_every=3000),
    "multi_block_fixed_mix": dict(level=1, strategy=zlib.Z_FIXED, sizes=[20000], flush_every=777),
    "max_member": dict(level=6, sizes=[65536]),
    "crc_lengths": dict(level=1, sizes=[4096, 4095, 4097, 63, 64, 65, 8192, 12345, 3, 61441]),
}


@pytest.mark.parametrize("shape", sorted(SHAPES))
@pytest.mark.parametrize("variant", ["default", "no_park", "tiles_of_2", "no_crc"])
def test_inflate_shapes(raw_bam, shape, variant, monkeypatch):
    # (huffman_only needs more token pages than the library's pool holds for it: the second-chance path with a worst-case pool)
    env = {"default": {}, "no_park": {"NGSQC_P1_PARK": "0"}, "tiles_of_2": {"NGSQC_TILE_MEMBERS": "2"}, "no_crc": {"NGSQC_VERIFY_CRC": "0"}}[variant]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    kw = dict(SHAPES[shape]); sizes = kw.pop("sizes")
    n = _roundtrip(raw_bam, rebgzf(raw_bam, sizes, **kw))
    assert n == 30000


@pytest.mark.parametrize("shape", ["level6_ragged", "huffman_only", "stored_small"])
def test_token_pool_that_runs_dry_in_every_tile(raw_bam, shape, monkeypatch):
    """a token pool of 1 % of the library's size runs out of pages in every chunk (literal-heavy, badly compressing members): the members that could
    not finish report K1_ERR_TOKEN_OVERFLOW and are inflated again, in batches, with a worst-case pool - in every tile of the stream; same bytes, same records"""
    monkeypatch.setenv("NGSQC_TILE_MEMBERS", "8"); monkeypatch.setenv("NGSQC_TOKEN_POOL_FACTOR", "0.01")
    kw = dict(SHAPES[shape]); sizes = kw.pop("sizes")
    assert _roundtrip(raw_bam, rebgzf(raw_bam, sizes, **kw)) == 30000


def test_runs_and_periodic_matches():
    # synthetic payload behind a valid BAM header: long runs (distance 1), short periods, far matches (distance ~32 KiB)
    rng = np.random.default_rng(3)
    hdr = b"BAM\x01" + struct.pack("<i", 0) + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chr1\x00" + struct.pack("<i", 1000000)
    blob = bytearray()
    base = rng.integers(0, 256, 40000, dtype=np.uint8).tobytes()
    blob += b"\x00" * 5000 + b"ab" * 3000 + b"abc" * 2000 + base + base[:32000] + b"\xff" * 70000 + base[100:20000]
    # wrap the blob as ONE oversized unmapped record so that the record chain is valid
    l_name, l_seq = 2, 0
    var = b"r\x00" + bytes(blob)
    rec = struct.pack("<iiBBHHHiiii", -1, -1, l_name, 0, 4680, 0, 4, l_seq, -1, -1, 0) + var
    raw = hdr + struct.pack("<i", len(rec)) + rec
    for kw in (dict(level=6), dict(level=9), dict(level=6, strategy=zlib.Z_RLE), dict(level=1)):
        assert _roundtrip(raw, rebgzf(raw, [65280, 30000, 65000], **kw)) == 1


def _flip_payload_byte(image: bytes, member_index: int, at: int) -> bytes:
    """flip one bit of the DEFLATE payload of one member (stored members: the byte is a literal of the inflated stream)"""
    img = bytearray(image); pos, k = 0, 0
    while pos < len(img):
        bs = struct.unpack_from("<H", img, pos + 16)[0] + 1
        if k == member_index:
            img[pos + 18 + 5 + at] ^= 0x10   # 5 = stored-block header (1 + 2 + 2 bytes)
            return bytes(img)
        pos += bs; k += 1
    raise AssertionError("member not found")


def test_crc32_mismatch_is_the_reference_read_error(raw_bam, monkeypatch):
    # stored members: a flipped payload byte leaves a valid stream of the right length; only the CRC32 can tell
    good = rebgzf(raw_bam, [60000], level=0)
    # pick a byte inside the SEQ field of a record that lies in member 3 (so that the record chain stays intact)
    o = 4; o += 4 + struct.unpack_from("<i", raw_bam, o)[0]; n_ref = struct.unpack_from("<i", raw_bam, o)[0]; o += 4
    for _ in range(n_ref):
        o += 4 + struct.unpack_from("<i", raw_bam, o)[0] + 4
    M = 40   # (behind the members that ngsqc_open inflates for the BAM header)
    while o < M * 60000 + 1000:
        o += 4 + struct.unpack_from("<i", raw_bam, o)[0]
    l_name, n_cig = raw_bam[o + 12], struct.unpack_from("<H", raw_bam, o + 16)[0]
    target = o + 36 + l_name + 4 * n_cig + 5
    assert M * 60000 <= target < (M + 1) * 60000
    bad = _flip_payload_byte(good, M, target - M * 60000)
    assert _roundtrip(raw_bam, good) == 30000
    h = ngsqc.Handle(data=np.frombuffer(bad, dtype=np.uint8))
    with pytest.raises(ngsqc.NgsqcError) as ei:
        h.decode()
    assert "Could not read next alignment" in str(ei.value) and f"CRC32 mismatch in block {M}" in str(ei.value) and ei.value.code == -2
    h.close()
    # switched off, the damaged byte goes through (what round 1 did)
    monkeypatch.setenv("NGSQC_VERIFY_CRC", "0")
    h = ngsqc.Handle(data=np.frombuffer(bad, dtype=np.uint8))
    got = h.inflated()
    assert got.size == len(raw_bam) and int((got != np.frombuffer(raw_bam, dtype=np.uint8)).sum()) == 1
    h.close()


CRC_SIZES = [1, 3, 15, 16, 17, 63, 64, 65, 4095, 4096, 4097, 8191, 8192, 8193, 12345, 16383, 16384, 16385, 20000, 32768, 40001, 49152, 61441, 65279, 65280, 65281, 65535, 65536]


@pytest.mark.parametrize("chains", ["1", "2", "4"])
def test_crc32_chains_per_lane(raw_bam, chains, monkeypatch):
    """the CRC kernel with 1 / 2 / 4 chains per lane (csrc/crc.hip; rounds of 4 / 8 / 16 KiB): member sizes on both sides of every round and piece boundary pass
    with their true CRC32, and a flipped bit anywhere in a member - first byte, last byte, the bytes around the boundaries - is found in exactly that member"""
    monkeypatch.setenv("NGSQC_CRC_CHAINS", chains)
    stored = sorted({min(n, 65500) for n in CRC_SIZES})   # (what fits a BGZF member as stored blocks: zlib ends a level-0 stream with an empty block of its own)
    good = rebgzf(raw_bam, stored, level=0)
    assert _roundtrip(raw_bam, good) == 30000
    assert _roundtrip(raw_bam, rebgzf(raw_bam, CRC_SIZES, level=6)) == 30000
    first_record = 4 + 4 + struct.unpack_from("<i", raw_bam, 4)[0]   # (damage in the header would be another error)
    n_ref = struct.unpack_from("<i", raw_bam, first_record)[0]; first_record += 4
    for _ in range(n_ref):
        first_record += 4 + struct.unpack_from("<i", raw_bam, first_record)[0] + 4
    starts, pos, k = [], 0, 0
    while pos < len(raw_bam):
        starts.append((pos, min(stored[k % len(stored)], len(raw_bam) - pos))); pos += starts[-1][1]; k += 1
    rng = np.random.default_rng(int(chains))
    picks = [m for m in range(len(starts)) if starts[m][0] > first_record + 200000][:3 * len(stored)]   # every size three times, behind the members an open inflates
    assert len(picks) == 3 * len(stored)
    for i, m in enumerate(picks):
        n = starts[m][1]
        at = [0, n - 1, int(rng.integers(0, n))][i // len(stored)]
        bad = _flip_payload_byte(good, m, at)
        with pytest.raises(ngsqc.NgsqcError) as ei:   # (the small members at the front are among those an open inflates for the header: the error may come from there)
            h = ngsqc.Handle(data=np.frombuffer(bad, dtype=np.uint8))
            try:
                h.decode()
            finally:
                h.close()
        assert f"CRC32 mismatch in block {m}" in str(ei.value) and ei.value.code == -2, (m, n, at, str(ei.value))


@pytest.mark.parametrize("field", ["l_seq", "n_cigar", "l_read_name", "neg_l_seq"])
def test_record_with_impossible_lengths_is_the_reference_read_error(raw_bam, field):
    """htslib's bam_read1 rejects a record whose variable-length fields do not fit block_size (the reference then throws "Could not read next
    alignment"); the kernels behind K2 trust l_read_name / n_cigar_op / l_seq, so K2 must refuse such a record instead of reading past it."""
    raw = bytearray(raw_bam)
    o = 4; o += 4 + struct.unpack_from("<i", raw, o)[0]; n_ref = struct.unpack_from("<i", raw, o)[0]; o += 4
    for _ in range(n_ref):
        o += 4 + struct.unpack_from("<i", raw, o)[0] + 4
    for _ in range(5000):
        o += 4 + struct.unpack_from("<i", raw, o)[0]
    if field == "l_seq":
        struct.pack_into("<i", raw, o + 20, 1 << 20)
    elif field == "neg_l_seq":
        struct.pack_into("<i", raw, o + 20, -5)
    elif field == "n_cigar":
        struct.pack_into("<H", raw, o + 16, 60000)
    else:
        raw[o + 12] = 0
    h = ngsqc.Handle(data=np.frombuffer(rebgzf(bytes(raw), [65280]), dtype=np.uint8))
    with pytest.raises(ngsqc.NgsqcError) as ei:
        h.n_records
    assert "Could not read next alignment" in str(ei.value) and ei.value.code == -2
    h.close()


def test_empty_members_inside_file(raw_bam):
    body = rebgzf(raw_bam, [20000])
    members = []
    pos = 0
    while pos < len(body):
        bs = struct.unpack_from("<H", body, pos + 16)[0] + 1
        members.append(body[pos:pos + bs]); pos += bs
    mixed = b"".join(m + (EOF_MEMBER if i % 3 == 0 else b"") for i, m in enumerate(members))
    assert _roundtrip(raw_bam, mixed) == 30000


@pytest.mark.timeout(300)
@pytest.mark.parametrize("seed", range(6))
def test_corrupt_payload_is_an_error_not_a_hang(raw_bam, seed):
    image = bytearray(rebgzf(raw_bam, [65280]))
    rng = np.random.default_rng(seed)
    # damage the deflate payload of several members (never the BGZF headers: those are validated on the host)
    pos, k = 0, 0
    while pos < len(image) - len(EOF_MEMBER):
        bs = struct.unpack_from("<H", image, pos + 16)[0] + 1
        if k >= 2 and k % 2 == 0:
            lo, hi = pos + 18, pos + bs - 8
            for _ in range(1 + seed):
                image[int(rng.integers(lo, hi))] ^= int(rng.integers(1, 256))
        pos += bs; k += 1
    h = None
    try:
        h = ngsqc.Handle(data=np.frombuffer(bytes(image), dtype=np.uint8))
        h.decode()
        got = h.inflated()
        # (only reachable when the damage cancels out in the CRC32 as well)
        assert got.size == len(raw_bam)
    except ngsqc.NgsqcError as e:
        assert "inflate" in str(e).lower() or "record" in str(e).lower() or "bam" in str(e).lower(), str(e)
    finally:
        if h is not None:
            h.close()


def test_members_of_thousands_of_deflate_blocks_take_the_third_chance():
    """zlib flush markers every two bytes make members of ~1500 DEFLATE blocks, each with its literal table in the token pool: more than the launch's pool AND more than
    the second chance's worst-case pool (four words per output byte) hold - the third chance (k1_pool_pages_absolute, 32 members per batch) inflates them. VERDICT r03:
    this path had only run under the wave emulator."""
    raw = gzip.decompress(G.generate(2000, seed=9, threads=2).tobytes())
    image = rebgzf(raw, [3000], level=6, flush_every=2)
    h = ngsqc.Handle(data=np.frombuffer(image, dtype=np.uint8))
    try:
        h.decode()
        got = h.inflated()
        assert got.size == len(raw) and np.array_equal(got, np.frombuffer(raw, dtype=np.uint8))
        assert h.n_records == 2000
        t = h.timings()
        assert t["members_third_chance"] > 0 and t["members_second_chance"] >= t["members_third_chance"]
    finally:
        h.close()
