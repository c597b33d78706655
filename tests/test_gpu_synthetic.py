"""GPU parity on synthetic BAMs (sizes the oracle finishes in seconds): short-read WGS shape, unaligned BGZF members
(records straddling members), ONT-like long reads with CG-tag CIGARs, coverage tools with min_baseq."""
import os

import numpy as np
import pytest

import bamgen_lib as G
import hostprep as H
import oracle_lib as O
from conftest import RESOURCES

pytestmark = pytest.mark.gpu
ngsqc = __import__("importlib").import_module("ngs-bits_amd")
OMIM = os.path.join(RESOURCES, "hg38_440_omim_genes.bed")
SKIP = {O.COUNTER_NAMES.index("half_depth"), O.COUNTER_NAMES.index("bases_covered_half")}


def _check(tmp_path, name, bed, qc_mode, merge_mode, **gen):
    path = str(tmp_path / name)
    G.write(path, **gen)
    ob = O.Bam(path)
    h = ngsqc.Handle(path=path)
    assert h.n_records == ob.count
    assert np.array_equal(h.record_offsets(), ob.record_offsets())
    regs = None
    if bed:
        regs, _ = H.bed_regions(bed, h.refs, merge_mode)
    tx, ty = H.xy_tids(h.refs)
    counters, _ = h.scan_mapping(qc_mode, regions=regs, tid_x=tx, tid_y=ty, nonspecial=H.nonspecial(h.refs))
    exp = O.mapping(ob, qc_mode, bed, merge_bed=(merge_mode == 1))
    for i in range(len(counters)):
        if i not in SKIP:
            assert int(counters[i]) == int(exp.counters[i]), (i, int(counters[i]), int(exp.counters[i]))
    if regs:
        assert np.array_equal(h.depth(int(counters[26])), exp.depth)
    t = h.timings()
    assert t["scan_algorithmic_bytes"] == ob.inflated_size - ob.first_record_offset
    h.close()
    return ob


def test_wgs_short_reads(tmp_path):
    _check(tmp_path, "wgs.bam", OMIM, ngsqc.MODE_WGS, 3, n_reads=300_000, seed=1, start_pos=15_900_000)


def test_wgs_unaligned_members(tmp_path):
    """htsjdk-style BGZF: records straddle members -> exercises the guess + verify + repair path of K2."""
    _check(tmp_path, "unaligned.bam", OMIM, ngsqc.MODE_WGS, 3, n_reads=120_000, seed=2, aligned=False, start_pos=15_900_000)


def test_chrX_chrY_counts(tmp_path):
    _check(tmp_path, "xy.bam", None, ngsqc.MODE_NOROI, 0, n_reads=100_000, seed=3, first_contig=22, start_pos=156_000_000, depth=2.0)


@pytest.mark.parametrize("long_mode", [None, "1", "0"])
def test_long_reads_cg_tag(tmp_path, monkeypatch, long_mode):
    """long_mode 1: the long-read form of K2's fast path (round 5: an entry is a group of sixteen members, every start guessed by a workgroup, tiles that begin inside a carried record
    ride the walk too) whatever the first record's size; 0: never; None: by the file's first record"""
    if long_mode is not None:
        monkeypatch.setenv("NGSQC_LONG_READ_MODE", long_mode)
    _check(tmp_path, "ont.bam", OMIM, ngsqc.MODE_WGS, 3, n_reads=1500, seed=4, mode=1, depth=40.0, start_pos=15_900_000)
    if long_mode == "1":   # the fused job (mapping scan riding the walk + site pileup over its candidates) against the oracle, single tile and 5-member tiles
        p = str(tmp_path / "ont.bam"); ob = O.Bam(p)
        for tm_ in (None, "5", "40"):
            if tm_: monkeypatch.setenv("NGSQC_TILE_MEMBERS", tm_)
            h = ngsqc.Handle(path=p)
            regs, _ = H.bed_regions(OMIM, h.refs, 3); tx, ty = H.xy_tids(h.refs); sites = [(0, p_) for p_ in range(15_900_500, 17_500_000, 4999)]
            out = h.run_job(mapping=dict(mode=ngsqc.MODE_WGS, regions=regs, min_mapq=1, tid_x=tx, tid_y=ty, nonspecial=H.nonspecial(h.refs)), sites=sites, site_params=(1, 13, True))
            exp = O.mapping(ob, ngsqc.MODE_WGS, OMIM, merge_bed=False)
            assert all(int(out["counters"][i]) == int(exp.counters[i]) for i in range(len(exp.counters)) if i not in SKIP)
            exp_sites = O.site_pileup(ob, sites, 1, 13, True)
            assert np.array_equal(out["site_counts"][:, :6], exp_sites) and int(exp_sites.sum()) > 100
            t = h.timings()
            # (a tile falls back to the host-verified path when a guess inside a long record was wrong - exact either way; with groups of 4 members about one 5-member
            # tile in five of this file did: its second group, one member, began inside a record longer than a member)
            assert t["tiles_chain_on_device"] >= t["n_tiles"] - 1 and t["walkers_per_member"] == -16, t
            h.close()


def test_roi_mode_exome_like(tmp_path):
    bed = tmp_path / "exome.bed"
    rng = np.random.default_rng(5)
    starts = np.sort(rng.integers(16_000_000, 17_400_000, 700))
    with open(bed, "w") as f:
        for s in starts:
            f.write(f"chr1\t{s}\t{s + int(rng.integers(60, 900))}\tex{s}\n")
    ob = _check(tmp_path, "roi.bam", str(bed), ngsqc.MODE_ROI, 1, n_reads=250_000, seed=6, start_pos=15_900_000)
    # coverage tools on the same pair, with and without base-quality mask
    h = ngsqc.Handle(path=str(tmp_path / "roi.bam"))
    regs, _ = H.bed_regions(str(bed), h.refs, 2)
    for baseq in (0, 20):
        h.scan_depth(regs, min_mapq=1, min_baseq=baseq)
        exp = O.low_high_coverage(ob, str(bed), 20, 1, baseq, is_high=False, random_access=True, tool_merge=1)
        assert np.array_equal(h.depth(exp["roi_bases"]), exp["depth"]), baseq
    lines, _ = H.bed_regions(str(bed), h.refs, 0)
    h.scan_depth(regs, min_mapq=1)
    cov, _, _ = O.avg_coverage(ob, str(bed), min_mapq=1, random_access=False)
    assert np.array_equal(h.region_sums(lines), cov)
    h.close()


@pytest.mark.parametrize("env", [{}, {"NGSQC_BASEQ_RIDE": "0"}, {"NGSQC_BQ_LIST_CAP": "7"}, {"NGSQC_TILE_MEMBERS": "11"},
                                 {"NGSQC_TILE_MEMBERS": "11", "NGSQC_BQ_LIST_CAP": "50"}, {"NGSQC_NO_FUSED_SCAN": "1"}])
def test_min_baseq_rides_the_walk(tmp_path, monkeypatch, env):
    """BedLowCoverage -min_baseq (BamAlignment::qualities): the depth scan rides K2's chain walk with min_baseq too (round 6: the default - the records that overlap a
    region go to a list, sorted by offset, and a lane per record counts their low-quality bases in an LDS tile of the difference array that leaves as one atomic per
    slot; NGSQC_BASEQ_RIDE=0: K2 + the thread-per-record scan). The same depth as the oracle either way, with a list that overflows (the tile is taken back and
    scanned record by record) and across tiles."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    bed = tmp_path / "x.bed"
    rng = np.random.default_rng(9)
    starts = np.sort(rng.integers(16_000_000, 16_900_000, 400))
    bed.write_text("".join(f"chr1\t{s_}\t{s_ + int(rng.integers(40, 700))}\n" for s_ in starts))
    p = str(tmp_path / "bq.bam"); G.write(p, n_reads=150_000, seed=21, start_pos=15_900_000)
    ob = O.Bam(p); h = ngsqc.Handle(path=p)
    regs, _ = H.bed_regions(str(bed), h.refs, 2)
    for baseq in (20, 30, 0):
        h.scan_depth(regs, min_mapq=1, min_baseq=baseq)
        exp = O.low_high_coverage(ob, str(bed), 20, 1, baseq, is_high=False, random_access=True, tool_merge=1)
        assert np.array_equal(h.depth(exp["roi_bases"]), exp["depth"]), baseq
        if baseq and "NGSQC_BASEQ_RIDE" not in env and "NGSQC_NO_FUSED_SCAN" not in env and "NGSQC_BQ_LIST_CAP" not in env:
            assert h.timings()["tiles_scan_fused"] == h.timings()["n_tiles"]
    h.close()


@pytest.mark.parametrize("tile_members", [1, 3, 7, 33])
def test_tiled_processing_matches_single_tile(tmp_path, monkeypatch, tile_members):
    """Files larger than HBM are processed in member ranges ("tiles"); records straddling tile borders are carried.
    Tiny tiles on unaligned / long-read inputs exercise every carry path; results must be identical."""
    monkeypatch.setenv("NGSQC_TILE_MEMBERS", str(tile_members))
    _check(tmp_path, "t_unaligned.bam", OMIM, ngsqc.MODE_WGS, 3, n_reads=30_000, seed=12, aligned=False, start_pos=15_900_000)
    _check(tmp_path, "t_ont.bam", OMIM, ngsqc.MODE_WGS, 3, n_reads=400, seed=13, mode=1, depth=40.0, start_pos=15_900_000)
    monkeypatch.setenv("NGSQC_LONG_READ_MODE", "1"); monkeypatch.setenv("NGSQC_GROUP_SHIFT", "2")     # groups of four members, tiles that cut groups and records
    _check(tmp_path, "t_ont.bam", OMIM, ngsqc.MODE_WGS, 3, n_reads=400, seed=13, mode=1, depth=40.0, start_pos=15_900_000)
    _check(tmp_path, "t_unaligned.bam", OMIM, ngsqc.MODE_WGS, 3, n_reads=30_000, seed=12, aligned=False, start_pos=15_900_000)
    monkeypatch.delenv("NGSQC_LONG_READ_MODE"); monkeypatch.delenv("NGSQC_GROUP_SHIFT")
    _check(tmp_path, "t_noroi.bam", None, ngsqc.MODE_NOROI, 0, n_reads=20_000, seed=14, first_contig=22, start_pos=156_000_000, depth=2.0)
    # inflated stream of a multi-tile file through the test hook
    path = str(tmp_path / "t_unaligned.bam")
    h = ngsqc.Handle(path=path)
    assert np.array_equal(h.inflated(), O.Bam(path).inflated())
    h.close()


@pytest.mark.parametrize("tile_members", [None, 5])
def test_fused_job_equals_single_purpose_calls(tmp_path, monkeypatch, tile_members):
    """ngsqc_run_job: mapping scan + extra depth scan + site pileup + raw-read QC in ONE pass over the BAM (every member inflated once) give exactly
    what the single-purpose entry points give in four passes - on a resident single-tile file and on a file streamed through many small tiles."""
    if tile_members:
        monkeypatch.setenv("NGSQC_TILE_MEMBERS", str(tile_members))
    path = str(tmp_path / "job.bam")
    G.write(path, n_reads=120_000, seed=61, aligned=False, start_pos=15_900_000)
    sub = tmp_path / "sub.bed"
    sub.write_text("chr1\t15950000\t15960000\nchr1\t16100000\t16100500\nchr1\t16300000\t16340000\n")
    ob = O.Bam(path)
    h = ngsqc.Handle(path=path)
    refs = h.refs
    regs, _ = H.bed_regions(OMIM, refs, 3)
    sub_regs, _ = H.bed_regions(str(sub), refs, 1)
    tx, ty = H.xy_tids(refs)
    sites = [(0, p_) for p_ in range(15_950_000, 16_350_000, 997)]
    mp = dict(mode=ngsqc.MODE_WGS, regions=regs, min_mapq=1, tid_x=tx, tid_y=ty, nonspecial=H.nonspecial(refs))
    out = h.run_job(mapping=mp, depth=dict(regions=sub_regs, min_mapq=1), sites=sites, site_params=(1, 13, False), read_qc=dict(single_end=False))
    tm = h.timings()
    assert tm["members_inflated"] == h.n_blocks and (tile_members is None or tm["n_tiles"] > 3)   # one K1 visit per member for all four consumers
    # depth set 0 = mapping ROI, set 1 = the extra depth scan
    roi_bases = int(out["counters"][26])
    d0 = h.depth(roi_bases)
    h.depth_select(1)
    d1 = h.depth(sum(e - s + 1 for _, s, e in sub_regs)); sums1 = h.region_sums(sub_regs)
    h.depth_select(0)
    assert np.array_equal(h.depth(roi_bases), d0)
    h.close()
    # the same through four single-purpose passes on a fresh handle
    g = ngsqc.Handle(path=path)
    c2, gc2 = g.scan_mapping(**mp)
    assert np.array_equal(out["counters"], c2) and np.array_equal(g.depth(roi_bases), d0)
    assert np.array_equal(g.site_pileup(sites, 1, 13, False), out["site_counts"])
    r2 = g.scan_reads(single_end=False)
    for k, v in r2.items():
        assert np.array_equal(np.asarray(out["reads"][k]), np.asarray(v)), k
    g.scan_depth(sub_regs, min_mapq=1)
    assert np.array_equal(g.depth(len(d1)), d1) and np.array_equal(g.region_sums(sub_regs), sums1)
    g.close()
    # and the oracle
    exp = O.mapping(ob, ngsqc.MODE_WGS, OMIM, merge_bed=False)
    for i in range(len(c2)):
        if i not in SKIP:
            assert int(out["counters"][i]) == int(exp.counters[i]), i
    assert np.array_equal(d0, exp.depth)
    assert np.array_equal(out["site_counts"][:, :6], O.site_pileup(ob, sites, 1, 13, False))
    cov, _, _ = O.avg_coverage(ob, str(sub), merge_bed=True, min_mapq=1, random_access=True)
    assert np.array_equal(sums1, cov) and int(d1.sum()) == int(cov.sum()) > 0


@pytest.mark.parametrize("aligned,tile_members", [(True, 0), (True, 9), (False, 9)])
def test_k2_variants_agree(tmp_path, monkeypatch, aligned, tile_members):
    """Every way to the record index gives the same job result: the scan riding K2's chain walk with one walker per BGZF member (default), with two, four
    and eight (NGSQC_WALKERS; pieces of a member whose first record is guessed, the chain checked on the device), compiled for four waves per SIMD
    (NGSQC_WALK_WAVES=4) and the plain K2 + scan (NGSQC_NO_FUSED_SCAN=1)."""
    p = str(tmp_path / "k2.bam")
    G.write(p, n_reads=50000, seed=77, aligned=aligned)
    if tile_members:
        monkeypatch.setenv("NGSQC_TILE_MEMBERS", str(tile_members))
    res = []
    for env in ({}, {"NGSQC_WALKERS": "2"}, {"NGSQC_WALKERS": "4"}, {"NGSQC_WALKERS": "8"}, {"NGSQC_WALK_WAVES": "4"}, {"NGSQC_NO_FUSED_SCAN": "1"}, {"NGSQC_NO_FUSED_SCAN": "1", "NGSQC_WALKERS": "4"},
                {"NGSQC_LONG_READ_MODE": "1"}, {"NGSQC_LONG_READ_MODE": "1", "NGSQC_GROUP_SHIFT": "1"}, {"NGSQC_EAGER_RECOFF": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        h = ngsqc.Handle(path=p)
        regs, _ = H.bed_regions(OMIM, h.refs, 3); tx, ty = H.xy_tids(h.refs)
        out = h.run_job(mapping=dict(mode=ngsqc.MODE_WGS, regions=regs, min_mapq=1, tid_x=tx, tid_y=ty, nonspecial=H.nonspecial(h.refs)), sites=H.known_sites(h.refs))
        res.append((out["counters"].copy(), out["site_counts"].copy(), h.depth(int(out["counters"][26])).copy()))
        tm = h.timings()
        if aligned and "NGSQC_NO_FUSED_SCAN" not in env and "NGSQC_LONG_READ_MODE" not in env:   # an htslib-style file: every tile's chain is checked on the device and scanned by the walk itself
            assert tm["tiles_chain_on_device"] == tm["n_tiles"] == tm["tiles_scan_fused"] and tm["walkers_per_member"] == int(env.get("NGSQC_WALKERS", 1)), tm
        h.close()
        for k in env:
            monkeypatch.delenv(k)
    for r in res[1:]:
        assert np.array_equal(r[0], res[0][0]) and np.array_equal(r[1], res[0][1]) and np.array_equal(r[2], res[0][2])
    assert res[0][0][0] > 0


def test_record_offsets_after_a_job_that_never_asked_for_them(tmp_path):
    """round 5: the fused MappingQC job works on the chain walk's own names and candidate lists - the record offsets of a tile are only expanded when a consumer
    reads them. A single-tile handle stays resident behind the job; whoever comes next (the BAI writer, a depth scan, the test hook) must still find them."""
    p = str(tmp_path / "lazy.bam")
    G.write(p, n_reads=30000, seed=5)
    h = ngsqc.Handle(path=p)
    regs, _ = H.bed_regions(OMIM, h.refs, 3); tx, ty = H.xy_tids(h.refs)
    h.run_job(mapping=dict(mode=ngsqc.MODE_WGS, regions=regs, min_mapq=1, tid_x=tx, tid_y=ty, nonspecial=H.nonspecial(h.refs)), sites=H.known_sites(h.refs))
    assert h.timings()["tiles_scan_fused"] == 1
    assert np.array_equal(h.record_offsets(), O.Bam(p).record_offsets())
    h.write_bai(str(tmp_path / "lazy.bai"))
    h2 = ngsqc.Handle(path=p); h2.write_bai(str(tmp_path / "eager.bai")); h2.close()
    assert open(str(tmp_path / "lazy.bai"), "rb").read() == open(str(tmp_path / "eager.bai"), "rb").read()
    h.close()


def _bgzf_member(payload):
    import struct, zlib
    c = zlib.compressobj(6, zlib.DEFLATED, -15); comp = c.compress(payload) + c.flush()
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload)))


def _record(name, pos, l_seq, qual, flag=0, mapq=60, tid=0):
    import struct
    nm = name + b"\0"; cigar = struct.pack("<I", (l_seq << 4) | 0)
    body = struct.pack("<iiBBHHHiiii", tid, pos, len(nm), mapq, 4680, 1, flag, l_seq, -1, -1, 0) + nm + cigar + bytes((l_seq + 1) // 2) + qual
    return struct.pack("<I", len(body)) + body


def test_false_start_guess_in_an_unaligned_tile(tmp_path):
    """a member that starts inside a record's qualities, at bytes that look like two chained record headers followed by one bam_read1 would refuse.
    The riding scan of K2's chain walk follows the false guess into the 'corrupt' record; the tile is not laid out like an htslib file, so the general path repairs
    the chain - and everything the riding scan added (the two fake reads included) must have been taken back: the counters are the oracle's, not twice that."""
    import struct
    text = b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:248956422\n"
    header = b"BAM\x01" + struct.pack("<I", len(text)) + text + struct.pack("<I", 1) + struct.pack("<I", 5) + b"chr1\0" + struct.pack("<I", 248956422)
    fake_ok = struct.pack("<IiiBBHHHiiii", 40, 0, 5, 1, 0, 0, 0, 0, 0, -1, -1, 0) + b"\0" + bytes(7)          # 44 bytes: a plausible record (name "", no CIGAR, no bases)
    fake_bad = struct.pack("<IiiBBHHHiiii", 40, 0, 5, 0, 0, 0, 0, 0, 0, -1, -1, 0) + bytes(8)                   # l_read_name 0: bam_read1 refuses it
    rng = np.random.default_rng(5)
    recs = [_record(b"r%04d" % i, 1000 + 10 * i, 100, bytes(rng.integers(2, 41, 100, dtype=np.uint8))) for i in range(60)]
    qual = bytearray(rng.integers(2, 41, 600, dtype=np.uint8)); q0 = 200
    qual[q0:q0 + 132] = fake_ok + fake_ok + fake_bad
    special = _record(b"special", 1700, 600, bytes(qual))
    tail = [_record(b"t%04d" % i, 1800 + 10 * i, 100, bytes(rng.integers(2, 41, 100, dtype=np.uint8))) for i in range(300)]
    stream = b"".join(recs) + special + b"".join(tail)
    cut = len(b"".join(recs)) + (len(special) - 600 + q0)      # the member border: the first fake header
    assert stream[cut:cut + 4] == struct.pack("<I", 40)
    members = [header, stream[:cut]]
    rest = stream[cut:]
    members += [rest[i:i + 20000] for i in range(0, len(rest), 20000)]
    path = str(tmp_path / "false_guess.bam")
    with open(path, "wb") as f:
        for m in members:
            f.write(_bgzf_member(m))
        f.write(_bgzf_member(b""))
    ob = O.Bam(path)
    assert ob.count == 361
    bed = str(tmp_path / "roi.bed")
    with open(bed, "w") as f:
        f.write("chr1\t900\t1500\nchr1\t1650\t1900\nchr1\t2500\t4000\n")
    for tiles in (None, "2"):
        if tiles:
            os.environ["NGSQC_TILE_MEMBERS"] = tiles
        try:
            h = ngsqc.Handle(path=path)
            regs, _ = H.bed_regions(bed, h.refs, 3)
            tx, ty = H.xy_tids(h.refs)
            out = h.run_job(mapping=dict(mode=ngsqc.MODE_WGS, regions=regs, min_mapq=1, tid_x=tx, tid_y=ty, nonspecial=H.nonspecial(h.refs)))
            exp = O.mapping(ob, ngsqc.MODE_WGS, bed, merge_bed=False)
            c = out["counters"]
            for i in range(len(c)):
                if i not in SKIP:
                    assert int(c[i]) == int(exp.counters[i]), (tiles, i, int(c[i]), int(exp.counters[i]))
            assert h.n_records == ob.count
            h.close()
        finally:
            os.environ.pop("NGSQC_TILE_MEMBERS", None)


def test_first_job_races_the_background_copy(tmp_path, monkeypatch):
    """ngsqc_open(path) returns while the compressed image is still on its way (member table walked in pieces by NGSQC_WALK_THREADS host threads - the default since
    round 4 - and copied by background threads); the first job starts at once and every K1 chunk waits for the pieces it reads. A slowed-down copy (1 MB pieces, a delay
    per piece) makes the chunks really wait; the counters must be those of a handle whose image was resident, and of the sequential member walk."""
    import time
    path = str(tmp_path / "race.bam")
    G.write(path, n_reads=300_000, seed=21, start_pos=15_900_000)

    def job(env):
        for k in ("NGSQC_WALK_THREADS", "NGSQC_H2D_PIECE_MB", "NGSQC_H2D_DELAY_US", "NGSQC_H2D_THREADS", "NGSQC_TILE_MEMBERS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        t0 = time.perf_counter(); h = ngsqc.Handle(path=path); t_open = time.perf_counter() - t0
        regs, _ = H.bed_regions(OMIM, h.refs, 3); tx, ty = H.xy_tids(h.refs)
        out = h.run_job(mapping=dict(mode=ngsqc.MODE_WGS, regions=regs, min_mapq=1, tid_x=tx, tid_y=ty, nonspecial=H.nonspecial(h.refs)))
        h.upload_wait(); t = h.timings(); n = h.n_records; h.close()
        return np.asarray(out["counters"]).copy(), t_open, t["h2d_ms"] * 1e-3, n

    ref, _, _, n_ref = job({"NGSQC_WALK_THREADS": "1"})
    assert n_ref == 300_000
    c8, _, _, _ = job({})                                   # the default: eight walkers
    assert np.array_equal(c8, ref)
    slow, t_open, t_h2d, _ = job({"NGSQC_H2D_PIECE_MB": "1", "NGSQC_H2D_DELAY_US": "4000", "NGSQC_H2D_THREADS": "2", "NGSQC_TILE_MEMBERS": "64"})
    assert np.array_equal(slow, ref)
    assert t_open < 0.5 * t_h2d, (t_open, t_h2d)            # open returned long before the last piece arrived: the job ran while the copy was in flight


@pytest.mark.parametrize("tile_members", ["64", "200"])
def test_streamed_image_equals_resident(tmp_path, monkeypatch, tile_members):
    """Round 4: ngsqc_open(path) of a large file keeps no resident compressed image - every job copies the file through a ring of K1-chunk slots (a slot is refilled when
    phase 2 of the chunk that used it is done). Forced here on a small file with tiny chunks (tens of chunks through four slots, slowed-down copies): same records, counters
    and depth as the resident handle, for a second job on the same handle (the file crosses PCIe again), for the BAI pass, and with the second-chance path in the stream."""
    path = str(tmp_path / "stream.bam")
    G.write(path, n_reads=200_000, seed=31, start_pos=15_900_000)
    ob = O.Bam(path)

    def run(env, jobs=1):
        for k in ("NGSQC_STREAM_IMAGE", "NGSQC_TILE_MEMBERS", "NGSQC_H2D_PIECE_MB", "NGSQC_H2D_DELAY_US", "NGSQC_TOKEN_POOL_FACTOR", "NGSQC_COMP_SLOTS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        h = ngsqc.Handle(path=path)
        regs, _ = H.bed_regions(OMIM, h.refs, 3); tx, ty = H.xy_tids(h.refs)
        res = []
        for _ in range(jobs):
            h.drop_decoded()
            out = h.run_job(mapping=dict(mode=ngsqc.MODE_WGS, regions=regs, min_mapq=1, tid_x=tx, tid_y=ty, nonspecial=H.nonspecial(h.refs)), sites=H.known_sites(h.refs))
            res.append((np.asarray(out["counters"]).copy(), np.asarray(out["site_counts"]).copy(), h.depth(int(out["counters"][26])).copy()))
        t = h.timings(); n = h.n_records; h.close()
        return res, t, n

    (ref,), _, n = run({"NGSQC_STREAM_IMAGE": "0", "NGSQC_TILE_MEMBERS": tile_members})
    assert n == ob.count
    res, t, n2 = run({"NGSQC_STREAM_IMAGE": "1", "NGSQC_TILE_MEMBERS": tile_members, "NGSQC_H2D_PIECE_MB": "1", "NGSQC_H2D_DELAY_US": "300"}, jobs=2)
    assert n2 == ob.count and t["members_inflated"] > 0
    for got in res:
        for a, b in zip(got, ref):
            assert np.array_equal(a, b)
    # two slots only, and a token pool that runs dry in every chunk: the second chance reads its members' bytes from the file, not from the ring
    res, t, _ = run({"NGSQC_STREAM_IMAGE": "1", "NGSQC_TILE_MEMBERS": tile_members, "NGSQC_COMP_SLOTS": "2", "NGSQC_TOKEN_POOL_FACTOR": "0.01"})
    assert t["members_second_chance"] > 0
    for a, b in zip(res[0], ref):
        assert np.array_equal(a, b)
    # the index pass is one more trip of the file through the ring
    monkeypatch.setenv("NGSQC_STREAM_IMAGE", "1"); monkeypatch.setenv("NGSQC_TILE_MEMBERS", tile_members); monkeypatch.delenv("NGSQC_TOKEN_POOL_FACTOR", raising=False)
    h = ngsqc.Handle(path=path)
    h.write_bai(str(tmp_path / "s.bai")); h.close()
    monkeypatch.setenv("NGSQC_STREAM_IMAGE", "0")
    h = ngsqc.Handle(path=path)
    h.write_bai(str(tmp_path / "r.bai")); h.close()
    assert open(str(tmp_path / "s.bai"), "rb").read() == open(str(tmp_path / "r.bai"), "rb").read()
