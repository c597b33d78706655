"""ctypes binding of oracle/liboracle.so (CPU restatement — the parity checker, never the product)."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
NCOUNTERS = 1032

COUNTER_NAMES = [
    "al_total", "al_mapped", "al_ontarget", "al_neartarget", "al_dup", "al_proper_paired", "insert_size_read_count",
    "bases_trimmed", "bases_mapped", "bases_clipped", "insert_size_sum", "bases_usable", "bases_usable_no_overlap",
    "bases_usable_raw", "bases_usable_roi", "bases_usable_dp0", "bases_usable_dp1", "bases_usable_dp2", "bases_usable_dp3",
    "bases_usable_dp4", "dp_dist0", "dp_dist1", "dp_dist2", "dp_dist3", "max_length", "paired_end", "roi_bases",
    "half_depth", "bases_covered_half", "reads_x", "reads_y", "yx_valid",
]

_lib = None


def build():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("capi.cpp", "stats.hpp", "stream.hpp", "bed.hpp", "bamio.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        vp, cp, i64, i32 = C.c_void_p, C.c_char_p, C.c_int64, C.c_int
        L.orc_bam_load.restype = vp; L.orc_bam_load.argtypes = [cp, cp, i32]
        L.orc_bam_free.argtypes = [vp]
        for f in ("orc_bam_count", "orc_bam_inflated_size", "orc_bam_first_record_offset", "orc_bam_n_blocks"):
            getattr(L, f).restype = i64; getattr(L, f).argtypes = [vp]
        L.orc_bam_sorted.restype = i32; L.orc_bam_sorted.argtypes = [vp]
        L.orc_bam_n_ref.restype = i32; L.orc_bam_n_ref.argtypes = [vp]
        L.orc_bam_ref_name.restype = cp; L.orc_bam_ref_name.argtypes = [vp, i32]
        L.orc_bam_ref_len.restype = i64; L.orc_bam_ref_len.argtypes = [vp, i32]
        L.orc_bam_inflated.restype = i64; L.orc_bam_inflated.argtypes = [vp, vp, i64]
        L.orc_bam_record_offsets.restype = i64; L.orc_bam_record_offsets.argtypes = [vp, vp, i64]
        L.orc_mapping.restype = vp; L.orc_mapping.argtypes = [vp, i32, cp, i32, cp, i32, i32, cp, i32]
        L.orc_result_counters.argtypes = [vp, vp]
        L.orc_result_text.restype = cp; L.orc_result_text.argtypes = [vp]
        L.orc_result_depth.restype = i64; L.orc_result_depth.argtypes = [vp, vp, i64]
        L.orc_result_gc.restype = i32; L.orc_result_gc.argtypes = [vp, vp, vp]
        L.orc_result_seconds.restype = C.c_double; L.orc_result_seconds.argtypes = [vp]
        L.orc_result_free.argtypes = [vp]
        L.orc_avg_coverage.restype = vp; L.orc_avg_coverage.argtypes = [vp, cp, i32, i32, i32, i32, i32, i32, cp, i32]
        L.orc_result_cov.restype = i64; L.orc_result_cov.argtypes = [vp, vp, i64]
        L.orc_result_bed.restype = cp; L.orc_result_bed.argtypes = [vp]
        L.orc_low_high_coverage.restype = vp; L.orc_low_high_coverage.argtypes = [vp, cp, i32, i32, i32, i32, i32, i32, cp, i32]
        L.orc_baseline_wgs_stream.restype = C.c_double
        L.orc_baseline_wgs_stream.argtypes = [vp, i64, cp, i32, i64, vp, vp, cp, i32]
        L.orc_bed_roundtrip.restype = i64; L.orc_bed_roundtrip.argtypes = [cp, i32, cp, i64, cp, i32]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


def _b(s):
    return None if s is None else os.fsencode(s)


class Bam:
    def __init__(self, path):
        err = C.create_string_buffer(1024)
        self.h = lib().orc_bam_load(_b(path), err, 1024)
        if not self.h:
            raise OracleError(err.value.decode())
        self.path = path

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_bam_free(self.h); self.h = None

    @property
    def count(self): return lib().orc_bam_count(self.h)
    @property
    def inflated_size(self): return lib().orc_bam_inflated_size(self.h)
    @property
    def first_record_offset(self): return lib().orc_bam_first_record_offset(self.h)
    @property
    def n_blocks(self): return lib().orc_bam_n_blocks(self.h)
    @property
    def refs(self):
        L = lib()
        return [(L.orc_bam_ref_name(self.h, i).decode(), L.orc_bam_ref_len(self.h, i)) for i in range(L.orc_bam_n_ref(self.h))]

    def inflated(self):
        a = np.empty(self.inflated_size, dtype=np.uint8)
        lib().orc_bam_inflated(self.h, a.ctypes.data, a.size)
        return a

    def record_offsets(self):
        a = np.empty(self.count, dtype=np.int64)
        lib().orc_bam_record_offsets(self.h, a.ctypes.data, a.size)
        return a


class MappingResult:
    def __init__(self, h):
        L = lib()
        self.counters = np.zeros(NCOUNTERS, dtype=np.int64)
        L.orc_result_counters(h, self.counters.ctypes.data)
        self.lines = []
        for ln in L.orc_result_text(h).decode().splitlines():
            acc, name, value, plot = ln.split("\t")
            self.lines.append((acc, name, value, plot == "1"))
        n = L.orc_result_depth(h, None, 0)
        self.depth = np.zeros(n, dtype=np.int32)
        if n:
            L.orc_result_depth(h, self.depth.ctypes.data, n)
        self.gc_roi = np.zeros(101); self.gc_reads = np.zeros(101)
        self.have_gc = bool(L.orc_result_gc(h, self.gc_roi.ctypes.data, self.gc_reads.ctypes.data))
        self.seconds = L.orc_result_seconds(h)
        L.orc_result_free(h)

    def __getitem__(self, name):
        return int(self.counters[COUNTER_NAMES.index(name)])

    @property
    def insert_hist(self):
        return self.counters[32:]

    def values(self):
        """name -> value string for non-plot lines (what MappingQC -txt prints)."""
        return {n: v for (_, n, v, p) in self.lines if not p}

    def txt(self):
        return [f"{n}: {v}" for (_, n, v, p) in self.lines if not p]


MODE_ROI, MODE_NOROI, MODE_WGS = 0, 1, 2


def mapping(bam, mode, bed=None, merge_bed=True, fasta=None, min_mapq=1, cfdna=False):
    err = C.create_string_buffer(1024)
    h = lib().orc_mapping(bam.h, mode, _b(bed), int(merge_bed), _b(fasta), min_mapq, int(cfdna), err, 1024)
    if not h:
        raise OracleError(err.value.decode())
    return MappingResult(h)


def avg_coverage(bam, bed, merge_bed=False, min_mapq=1, decimals=2, random_access=False, skip_mismapped=False, clear=False):
    err = C.create_string_buffer(1024)
    L = lib()
    h = L.orc_avg_coverage(bam.h, _b(bed), int(merge_bed), min_mapq, decimals, int(random_access), int(skip_mismapped), int(clear), err, 1024)
    if not h:
        raise OracleError(err.value.decode())
    n = L.orc_result_cov(h, None, 0)
    cov = np.zeros(n, dtype=np.int64)
    if n:
        L.orc_result_cov(h, cov.ctypes.data, n)
    text = L.orc_result_bed(h).decode()
    secs = L.orc_result_seconds(h)
    L.orc_result_free(h)
    return cov, text, secs


def read_counts(bam, bed, min_mapq=1):
    """BedReadCount: (counts per line of the merged BED, BED text with the count as annotation)."""
    err = C.create_string_buffer(1024)
    L = lib()
    L.orc_read_counts.restype = C.c_void_p; L.orc_read_counts.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    h = L.orc_read_counts(bam.h, _b(bed), min_mapq, err, 1024)
    if not h:
        raise OracleError(err.value.decode())
    n = L.orc_result_cov(h, None, 0)
    cov = np.zeros(max(n, 1), dtype=np.int64)
    L.orc_result_cov(h, cov.ctypes.data, n)
    text = L.orc_result_bed(h).decode()
    L.orc_result_free(h)
    return cov[:n], text


def low_high_coverage(bam, bed, cutoff, min_mapq=1, min_baseq=0, is_high=False, random_access=False, tool_merge=1):
    """tool_merge: 1 = merge(true,true) like the tools, 2 = plain merge() like the unit tests, 0 = none."""
    err = C.create_string_buffer(1024)
    L = lib()
    h = L.orc_low_high_coverage(bam.h, _b(bed), tool_merge, cutoff, min_mapq, min_baseq, int(is_high), int(random_access), err, 1024)
    if not h:
        raise OracleError(err.value.decode())
    st = np.zeros(4, dtype=np.int64)
    L.orc_result_cov(h, st.ctypes.data, 4)
    n = L.orc_result_depth(h, None, 0)
    depth = np.zeros(n, dtype=np.int32)
    if n:
        L.orc_result_depth(h, depth.ctypes.data, n)
    text = L.orc_result_bed(h).decode()
    secs = L.orc_result_seconds(h)
    L.orc_result_free(h)
    return {"roi_regions": int(st[0]), "roi_bases": int(st[1]), "out_regions": int(st[2]), "out_bases": int(st[3]),
            "bed": text, "depth": depth, "seconds": secs}


def bed_roundtrip(bed, merge_mode=0):
    err = C.create_string_buffer(1024)
    n = lib().orc_bed_roundtrip(_b(bed), merge_mode, None, 0, err, 1024)
    if n < 0:
        raise OracleError(err.value.decode())
    buf = C.create_string_buffer(n + 1)
    lib().orc_bed_roundtrip(_b(bed), merge_mode, buf, n + 1, err, 1024)
    return buf.value.decode()


def gc_bins(fasta, bed, merge_mode=1):
    """GC bin (0..100, or -1) of every chunk of roi.chunk(100), in chunk order (Statistics.cpp:363-387)."""
    L = lib()
    L.orc_gc_bins.restype = C.c_int64
    L.orc_gc_bins.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int64, C.c_char_p, C.c_int]
    err = C.create_string_buffer(1024)
    n = L.orc_gc_bins(_b(fasta), _b(bed), merge_mode, None, 0, err, 1024)
    if n < 0:
        raise OracleError(err.value.decode())
    out = np.zeros(max(n, 1), dtype=np.int32)
    L.orc_gc_bins(_b(fasta), _b(bed), merge_mode, out.ctypes.data, n, err, 1024)
    return out[:n]


def baseline_wgs_stream(image, bed=None, min_mapq=1, max_records=-1, sites=None, site_params=(1, 13, False)):
    """Streaming single-thread MappingQC -wgs loop on a BAM image (numpy uint8). Returns (counters, stats dict, seconds). sites: rows of (tid, 1-based pos, ...) - the
    contamination pileup of MappingQC's third pass rides the same loop (site_params = (min_mapq, min_baseq, include_not_properly_paired)); stats["site_counts"] = int64[n, 6]."""
    img = np.ascontiguousarray(image, dtype=np.uint8)
    counters = np.zeros(NCOUNTERS, dtype=np.int64)
    st = np.zeros(3, dtype=np.int64)
    err = C.create_string_buffer(1024)
    L = lib()
    if sites is None:
        secs = L.orc_baseline_wgs_stream(img.ctypes.data, img.size, _b(bed), min_mapq, max_records, counters.ctypes.data, st.ctypes.data, err, 1024)
        site_counts = None
    else:
        s2 = np.asarray(sites, dtype=np.int64).reshape(len(sites), -1)
        tid = np.ascontiguousarray(s2[:, 0], dtype=np.int32); pos = np.ascontiguousarray(s2[:, 1], dtype=np.int32)
        site_counts = np.zeros((len(tid), 6), dtype=np.int64)
        L.orc_baseline_wgs_stream_sites.restype = C.c_double
        L.orc_baseline_wgs_stream_sites.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_char_p, C.c_int]
        secs = L.orc_baseline_wgs_stream_sites(img.ctypes.data, img.size, _b(bed), min_mapq, max_records, tid.ctypes.data, pos.ctypes.data, len(tid), int(site_params[0]), int(site_params[1]),
                                               int(bool(site_params[2])), counters.ctypes.data, st.ctypes.data, site_counts.ctypes.data, err, 1024)
    if secs < 0:
        raise OracleError(err.value.decode())
    out = {"n_records": int(st[0]), "inflated": int(st[1]), "compressed": int(st[2])}
    if site_counts is not None:
        out["site_counts"] = site_counts
    return counters, out, secs


def site_pileup(bam, sites, min_mapq=1, min_baseq=13, include_not_properly_paired=False):
    """BamReader::getPileup SNP counts for (tid, pos) sites: int64[n, 6] = A, C, G, T, N, deletion."""
    n = len(sites)
    tid = np.ascontiguousarray([t for t, _ in sites], dtype=np.int32); pos = np.ascontiguousarray([p for _, p in sites], dtype=np.int32)
    out = np.zeros((max(n, 1), 6), dtype=np.int64)
    err = C.create_string_buffer(1024)
    L = lib()
    L.orc_site_pileup.restype = C.c_int
    L.orc_site_pileup.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_int]
    if L.orc_site_pileup(bam.h, tid.ctypes.data, pos.ctypes.data, n, min_mapq, int(include_not_properly_paired), min_baseq, out.ctypes.data, err, 1024) != 0:
        raise OracleError(err.value.decode())
    return out[:n]


def contamination(bam, snps, include_not_properly_paired=False):
    """Statistics::contamination on known SNVs [(tid, pos, ref, alt)] (already AF / SNV / ROI filtered). Returns the QC value string."""
    n = len(snps)
    tid = np.ascontiguousarray([s[0] for s in snps], dtype=np.int32); pos = np.ascontiguousarray([s[1] for s in snps], dtype=np.int32)
    ref = bytes(ord(s[2]) for s in snps); alt = bytes(ord(s[3]) for s in snps)
    out = C.create_string_buffer(64); err = C.create_string_buffer(1024)
    L = lib()
    L.orc_contamination.restype = C.c_int
    L.orc_contamination.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_int64, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    if L.orc_contamination(bam.h, tid.ctypes.data, pos.ctypes.data, ref, alt, n, int(include_not_properly_paired), out, 64, err, 1024) != 0:
        raise OracleError(err.value.decode())
    return out.value.decode()


def reads_qc(bam, single_end=False, len_cap=None, n_cycles=320):
    """StatisticsReads::update(BamAlignment) over the whole BAM: dict in the layout of ngs-bits_amd Handle.scan_reads()."""
    L = lib()
    L.orc_reads_qc.restype = C.c_int
    L.orc_reads_qc.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_char_p, C.c_int]
    out = np.zeros(333, dtype=np.int64); err = C.create_string_buffer(1024)
    if L.orc_reads_qc(bam.h, int(single_end), out.ctypes.data, None, 0, None, 0, err, 1024) != 0:
        raise OracleError(err.value.decode())
    cap = int(out[6]) if len_cap is None else len_cap
    lens = np.zeros(cap + 1, dtype=np.int64); cyc = np.zeros((max(n_cycles, 1), 7), dtype=np.int64)
    if L.orc_reads_qc(bam.h, int(single_end), out.ctypes.data, lens.ctypes.data, cap, cyc.ctypes.data, n_cycles, err, 1024) != 0:
        raise OracleError(err.value.decode())
    return dict(c_forward=int(out[0]), c_reverse=int(out[1]), bases_sequenced=int(out[2]), c_read_q20=int(out[3]), c_base_q20=int(out[4]),
                c_base_q30=int(out[5]), max_cycles=int(out[6]), base_qualities=out[8:108].copy(), read_qualities=out[108:208].copy(),
                qscore_dist_r1=out[208:268].copy(), qscore_dist_r2=out[268:328].copy(), bases=out[328:333].copy(), read_lengths=lens, cycles=cyc[:n_cycles])


ORDER_DEPENDENT = (7, 12, 24, 25, 26, 27, 28, 31)   # counters the all-cores baseline cannot sum (see oracle/capi.cpp)


def baseline_wgs_stream_mt(image, bed=None, min_mapq=1, threads=0, want_counters=False, hist_cap=599):
    """All-cores form of the bench baseline. Returns (stats dict, seconds) or, with want_counters, (stats, seconds, summed counters,
    depth histogram): the additive counters (every index not in ORDER_DEPENDENT) and the histogram are exact for an aligned BAM."""
    import os as _os
    L = lib()
    L.orc_baseline_wgs_stream_mt.restype = C.c_double
    L.orc_baseline_wgs_stream_mt.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    buf = np.ascontiguousarray(np.frombuffer(image, dtype=np.uint8))
    st = np.zeros(3, dtype=np.int64); err = C.create_string_buffer(1024)
    counters = np.zeros(NCOUNTERS, dtype=np.int64) if want_counters else None
    hist = np.zeros(hist_cap + 1, dtype=np.int64) if want_counters else None
    secs = L.orc_baseline_wgs_stream_mt(buf.ctypes.data, buf.size, _b(bed), min_mapq, threads or (_os.cpu_count() or 1), st.ctypes.data,
                                         counters.ctypes.data if want_counters else None, hist.ctypes.data if want_counters else None, hist_cap, err, 1024)
    if secs < 0:
        raise OracleError(err.value.decode())
    stats = dict(n_records=int(st[0]), inflated=int(st[1]), compressed=int(st[2]))
    return (stats, secs, counters, hist) if want_counters else (stats, secs)
