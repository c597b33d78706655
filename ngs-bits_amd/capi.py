"""ctypes binding of include/ngsqc.h (libngsqc_hip.so). Plumbing only — all compute is in the HIP library."""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
MODE_ROI, MODE_NOROI, MODE_WGS = 0, 1, 2
NCOUNTERS = 1032
COUNTER_NAMES = [
    "al_total", "al_mapped", "al_ontarget", "al_neartarget", "al_dup", "al_proper_paired", "insert_size_read_count",
    "bases_trimmed", "bases_mapped", "bases_clipped", "insert_size_sum", "bases_usable", "bases_usable_no_overlap",
    "bases_usable_raw", "bases_usable_roi", "bases_usable_dp0", "bases_usable_dp1", "bases_usable_dp2", "bases_usable_dp3",
    "bases_usable_dp4", "dp_dist0", "dp_dist1", "dp_dist2", "dp_dist3", "max_length", "paired_end", "roi_bases",
    "half_depth", "bases_covered_half", "reads_x", "reads_y", "yx_valid",
]


class NgsqcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ngsqc error {code}: {msg}")
        self.code = code
        self.message = msg


class Region(C.Structure):
    _fields_ = [("tid", C.c_int32), ("start", C.c_int32), ("end", C.c_int32)]


class Run(C.Structure):
    _fields_ = [("line", C.c_int64), ("start", C.c_int32), ("end", C.c_int32)]


class MappingParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("min_mapq", C.c_int32), ("tid_x", C.c_int32), ("tid_y", C.c_int32),
                ("tid_nonspecial", C.c_void_p), ("regions", C.c_void_p), ("n_regions", C.c_int64),
                ("gc_chunks", C.c_void_p), ("gc_bin", C.c_void_p), ("n_gc_chunks", C.c_int64)]


class DepthParams(C.Structure):
    _fields_ = [("min_mapq", C.c_int32), ("min_baseq", C.c_int32), ("skip_mismapped", C.c_int32), ("reserved", C.c_int32),
                ("regions", C.c_void_p), ("n_regions", C.c_int64)]


class Timings(C.Structure):
    _fields_ = [("h2d_ms", C.c_double), ("inflate_ms", C.c_double), ("index_ms", C.c_double), ("scan_ms", C.c_double),
                ("finalize_ms", C.c_double), ("total_ms", C.c_double), ("inflate_launches", C.c_int64),
                ("scan_launches", C.c_int64), ("scan_algorithmic_bytes", C.c_int64), ("compressed_bytes", C.c_int64),
                ("inflated_bytes", C.c_int64), ("n_records", C.c_int64), ("scan_kernel_ms", C.c_double), ("depth_kernel_ms", C.c_double), ("inflate_huff_ms", C.c_double), ("inflate_lz77_ms", C.c_double),
                ("inflate_huff_launches", C.c_int64), ("n_tiles", C.c_int64), ("members_inflated", C.c_int64), ("depth_scan_ms", C.c_double),
                ("pileup_ms", C.c_double), ("reads_ms", C.c_double), ("job_wall_ms", C.c_double),
                ("members_second_chance", C.c_int64), ("members_third_chance", C.c_int64),
                ("tiles_chain_on_device", C.c_int64), ("tiles_scan_fused", C.c_int64), ("walkers_per_member", C.c_int64), ("switches", C.c_int64)]


SW_VERIFY_CRC, SW_CRAM_IGNORE_MD5, SW_CRAM_NO_REFERENCE = 1, 2, 4   # Timings.switches: the switches that can change a result (default: SW_VERIFY_CRC alone)


class JobDesc(C.Structure):
    _fields_ = [("mapping", C.c_void_p), ("depth", C.c_void_p), ("sites", C.c_void_p), ("n_sites", C.c_int64),
                ("site_min_mapq", C.c_int32), ("site_min_baseq", C.c_int32), ("site_include_npp", C.c_int32),
                ("read_qc", C.c_int32), ("read_qc_single_end", C.c_int32), ("reserved", C.c_int32)]


class JobResult(C.Structure):
    _fields_ = [("counters", C.c_void_p), ("gc_reads", C.c_void_p), ("site_counts", C.c_void_p), ("read_stats", C.c_void_p)]


class ShardSummary(C.Structure):
    _fields_ = [("n_records", C.c_int64), ("first_abs", C.c_int64), ("exit_abs", C.c_int64), ("max_len", C.c_int64),
                ("first_max_ord", C.c_int64), ("first_paired_ord", C.c_int64)]


class ShardFix(C.Structure):
    _fields_ = [("gmax", C.c_int64), ("floor_max", C.c_int64), ("trim_upto", C.c_int64), ("paired_upto", C.c_int64),
                ("paired_end", C.c_int32), ("reserved", C.c_int32)]


SUMMARY_FIELDS = [f for f, _ in ShardSummary._fields_]


class ReadStats(C.Structure):
    _fields_ = [("c_forward", C.c_int64), ("c_reverse", C.c_int64), ("bases_sequenced", C.c_int64), ("bases", C.c_int64 * 5),
                ("base_qualities", C.c_int64 * 100), ("read_qualities", C.c_int64 * 100), ("qscore_dist_r1", C.c_int64 * 60),
                ("qscore_dist_r2", C.c_int64 * 60), ("max_cycles", C.c_int64), ("n_unknown_base", C.c_int64), ("n_quality_out_of_range", C.c_int64)]


def lib_path():
    return os.path.join(PKG_DIR, "libngsqc_hip.so")


def build_library(force=False):
    """hipcc --offload-arch=gfx950 build of csrc/ (cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(PKG_DIR, "csrc"), "-s", "-j8"]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args)
    return lib_path()


_lib = None


def lib():
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise NgsqcError(-4, f"{p} is missing: build it with __graft_entry__.build() (there is no CPU fallback)")
        L = C.CDLL(p)
        vp, cp, i64, i32 = C.c_void_p, C.c_char_p, C.c_int64, C.c_int
        L.ngsqc_open.restype = i32; L.ngsqc_open.argtypes = [cp, i32, C.POINTER(vp)]
        L.ngsqc_open_memory.restype = i32; L.ngsqc_open_memory.argtypes = [vp, C.c_size_t, i32, C.POINTER(vp)]
        L.ngsqc_close.argtypes = [vp]
        L.ngsqc_last_error.restype = cp; L.ngsqc_last_error.argtypes = [vp]
        L.ngsqc_n_ref.restype = i32; L.ngsqc_n_ref.argtypes = [vp]
        L.ngsqc_ref_name.restype = cp; L.ngsqc_ref_name.argtypes = [vp, i32]
        L.ngsqc_ref_len.restype = i64; L.ngsqc_ref_len.argtypes = [vp, i32]
        for f in ("ngsqc_n_records", "ngsqc_inflated_size", "ngsqc_n_bgzf_blocks", "ngsqc_compressed_size"):
            getattr(L, f).restype = i64; getattr(L, f).argtypes = [vp]
        for f in ("ngsqc_decode", "ngsqc_drop_decoded"):
            getattr(L, f).restype = i32; getattr(L, f).argtypes = [vp]
        L.ngsqc_copy_inflated.restype = i32; L.ngsqc_copy_inflated.argtypes = [vp, vp, i64]
        L.ngsqc_copy_record_offsets.restype = i32; L.ngsqc_copy_record_offsets.argtypes = [vp, vp, i64]
        L.ngsqc_scan_mapping.restype = i32; L.ngsqc_scan_mapping.argtypes = [vp, C.POINTER(MappingParams), vp, vp]
        L.ngsqc_scan_depth.restype = i32; L.ngsqc_scan_depth.argtypes = [vp, C.POINTER(DepthParams)]
        L.ngsqc_scan_depth_partial.restype = i32; L.ngsqc_scan_depth_partial.argtypes = [vp, C.POINTER(DepthParams)]
        L.ngsqc_depth_stats.restype = i32; L.ngsqc_depth_stats.argtypes = [vp, C.c_int32, i64, vp, vp]
        L.ngsqc_depth_copy.restype = i32; L.ngsqc_depth_copy.argtypes = [vp, vp, i64]
        L.ngsqc_region_sums.restype = i32; L.ngsqc_region_sums.argtypes = [vp, vp, i64, vp]
        L.ngsqc_lowhigh_runs.restype = i32
        L.ngsqc_lowhigh_runs.argtypes = [vp, vp, i64, C.c_int32, C.c_int32, C.c_int32, vp, i64, C.POINTER(i64)]
        L.ngsqc_get_timings.restype = i32; L.ngsqc_get_timings.argtypes = [vp, C.POINTER(Timings)]
        L.ngsqc_get_timings_sized.restype = i32; L.ngsqc_get_timings_sized.argtypes = [vp, vp, C.c_size_t]; L.ngsqc_abi_version.restype = i32
        L.ngsqc_version.restype = cp; L.ngsqc_device_count.restype = i32; L.ngsqc_device_count.argtypes = []
        L.ngsqc_site_pileup.restype = i32; L.ngsqc_site_pileup.argtypes = [vp, vp, i64, C.c_int32, C.c_int32, C.c_int32, vp]
        L.ngsqc_scan_reads.restype = i32; L.ngsqc_scan_reads.argtypes = [vp, C.c_int32, C.POINTER(ReadStats)]
        L.ngsqc_read_length_hist.restype = i32; L.ngsqc_read_length_hist.argtypes = [vp, vp, i64]
        L.ngsqc_read_cycle_stats.restype = i32; L.ngsqc_read_cycle_stats.argtypes = [vp, vp, i64]
        L.ngsqc_open_shard.restype = i32; L.ngsqc_open_shard.argtypes = [cp, i32, i32, i32, C.POINTER(vp)]
        L.ngsqc_open_memory_shard.restype = i32; L.ngsqc_open_memory_shard.argtypes = [vp, C.c_size_t, i32, i32, i32, C.POINTER(vp)]
        L.ngsqc_comm_unique_id.restype = i32; L.ngsqc_comm_unique_id.argtypes = [vp]
        L.ngsqc_comm_init.restype = i32; L.ngsqc_comm_init.argtypes = [i32, i32, vp, i32, C.POINTER(vp)]
        L.ngsqc_comm_destroy.restype = i32; L.ngsqc_comm_destroy.argtypes = [vp]
        L.ngsqc_comm_last_error.restype = cp; L.ngsqc_comm_last_error.argtypes = [vp]
        L.ngsqc_comm_allreduce_counters.restype = i32; L.ngsqc_comm_allreduce_counters.argtypes = [vp, vp]
        L.ngsqc_comm_allreduce_i64.restype = i32; L.ngsqc_comm_allreduce_i64.argtypes = [vp, vp, i64, i32]
        L.ngsqc_comm_allreduce_f64.restype = i32; L.ngsqc_comm_allreduce_f64.argtypes = [vp, vp, i64]
        L.ngsqc_comm_allgather_summaries.restype = i32; L.ngsqc_comm_allgather_summaries.argtypes = [vp, C.POINTER(ShardSummary), C.POINTER(ShardSummary)]
        L.ngsqc_comm_allreduce_depth.restype = i32; L.ngsqc_comm_allreduce_depth.argtypes = [vp, vp]
        L.ngsqc_scan_mapping_partial.restype = i32; L.ngsqc_scan_mapping_partial.argtypes = [vp, C.POINTER(MappingParams), C.POINTER(ShardSummary)]
        L.ngsqc_plan_shard_fix.restype = i32; L.ngsqc_plan_shard_fix.argtypes = [C.POINTER(ShardSummary), i32, i32, C.POINTER(ShardFix)]
        L.ngsqc_scan_mapping_finish.restype = i32; L.ngsqc_scan_mapping_finish.argtypes = [vp, C.POINTER(ShardFix), vp, vp]
        L.ngsqc_depth_device.restype = i32; L.ngsqc_depth_device.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
        L.ngsqc_depth_diff_copy.restype = i32; L.ngsqc_depth_diff_copy.argtypes = [vp, vp, i64]
        L.ngsqc_depth_diff_set.restype = i32; L.ngsqc_depth_diff_set.argtypes = [vp, vp, i64]
        L.ngsqc_depth_finalize.restype = i32; L.ngsqc_depth_finalize.argtypes = [vp]
        L.ngsqc_region_read_counts.restype = i32; L.ngsqc_region_read_counts.argtypes = [vp, vp, i64, C.c_int32, vp]
        L.ngsqc_run_job.restype = i32; L.ngsqc_run_job.argtypes = [vp, C.POINTER(JobDesc), C.POINTER(JobResult)]
        L.ngsqc_depth_select.restype = i32; L.ngsqc_depth_select.argtypes = [vp, C.c_int32]
        L.ngsqc_depth_reduce.restype = i32; L.ngsqc_depth_reduce.argtypes = [vp, C.POINTER(vp), i32]
        _lib = L
    return _lib


def device_count():
    """HIP devices the library's own runtime sees (include/ngsqc.h ngsqc_device_count) - not torch's: a second HIP / HSA runtime in the process is what a test that
    only wants a number must not load."""
    return int(lib().ngsqc_device_count())


EXPORTS = [
    "ngsqc_open", "ngsqc_open_memory", "ngsqc_close", "ngsqc_last_error", "ngsqc_n_ref", "ngsqc_ref_name", "ngsqc_ref_len",
    "ngsqc_n_records", "ngsqc_inflated_size", "ngsqc_n_bgzf_blocks", "ngsqc_compressed_size", "ngsqc_decode",
    "ngsqc_drop_decoded", "ngsqc_copy_inflated", "ngsqc_copy_record_offsets", "ngsqc_scan_mapping", "ngsqc_scan_depth",
    "ngsqc_depth_stats", "ngsqc_depth_copy", "ngsqc_region_sums", "ngsqc_lowhigh_runs", "ngsqc_get_timings", "ngsqc_get_timings_sized", "ngsqc_abi_version", "ngsqc_version", "ngsqc_device_count",
    "ngsqc_site_pileup", "ngsqc_scan_reads", "ngsqc_read_length_hist", "ngsqc_read_cycle_stats", "ngsqc_open_shard", "ngsqc_open_memory_shard", "ngsqc_scan_mapping_partial", "ngsqc_scan_depth_partial", "ngsqc_plan_shard_fix", "ngsqc_scan_mapping_finish",
    "ngsqc_depth_device", "ngsqc_depth_diff_copy", "ngsqc_depth_diff_set", "ngsqc_depth_finalize",
    "ngsqc_run_job", "ngsqc_depth_select", "ngsqc_depth_reduce", "ngsqc_region_read_counts", "ngsqc_upload_wait", "ngsqc_run_job_partial", "ngsqc_bai_range", "ngsqc_open_range", "ngsqc_header_text", "ngsqc_open_regions", "ngsqc_open_head",
    "ngsqc_write_bai", "ngsqc_bai_assemble", "ngsqc_bgzf_scan", "ngsqc_write_csi", "ngsqc_csi_assemble", "ngsqc_bai_ranges",
    "ngsqc_set_reference", "ngsqc_set_cram_skip", "ngsqc_set_cram_skip_thread", "ngsqc_cram_to_bam",
]


def bai_range(bam_path, regions, n_ref):
    """regions: [(tid, start, end)] 1-based closed. Returns (beg_voff, end_voff, found) from <bam>.bai (host only); raises NgsqcError without an index."""
    L = lib()
    L.ngsqc_bai_range.restype = C.c_int
    L.ngsqc_bai_range.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
    arr = _regions_array(regions); b, e, f = C.c_uint64(0), C.c_uint64(0), C.c_int32(0)
    rc = L.ngsqc_bai_range(os.fsencode(bam_path), C.cast(arr, C.c_void_p), len(regions), int(n_ref), C.byref(b), C.byref(e), C.byref(f))
    if rc != 0:
        raise NgsqcError(rc, L.ngsqc_last_error(None).decode("utf-8", "replace"))
    return int(b.value), int(e.value), bool(f.value)


def bai_ranges(bam_path, regions, n_ref):
    """Per-region virtual-offset ranges from <bam>.bai (host only, one load of the index): [(beg_voff, end_voff)], end_voff == 0 when no record can overlap."""
    L = lib()
    L.ngsqc_bai_ranges.restype = C.c_int
    L.ngsqc_bai_ranges.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    arr = _regions_array(regions); n = len(regions)
    b = (C.c_uint64 * max(n, 1))(); e = (C.c_uint64 * max(n, 1))()
    rc = L.ngsqc_bai_ranges(os.fsencode(bam_path), C.cast(arr, C.c_void_p), n, int(n_ref), b, e)
    if rc != 0:
        raise NgsqcError(rc, L.ngsqc_last_error(None).decode("utf-8", "replace"))
    return [(int(b[i]), int(e[i])) for i in range(n)]


def set_reference(fasta_path):
    """The reference genome (FASTA with .fai) CRAM files are decoded against (process-wide, like the reference's RefGenomeService); None: none."""
    L = lib(); L.ngsqc_set_reference.restype = C.c_int; L.ngsqc_set_reference.argtypes = [C.c_char_p]
    L.ngsqc_set_reference(os.fsencode(fasta_path) if fasta_path else None)


CRAM_SKIP_NAMES, CRAM_SKIP_TAGS = 1, 2


def set_cram_skip(flags):
    """What later opens of a CRAM need not decode (ngsqc_set_cram_skip: read names and / or optional fields; BamReader::skipTags in the reference)."""
    L = lib(); L.ngsqc_set_cram_skip.restype = C.c_int; L.ngsqc_set_cram_skip.argtypes = [C.c_int32]
    if L.ngsqc_set_cram_skip(int(flags)) != 0:
        raise ValueError("invalid CRAM skip flags")


def set_cram_skip_thread(flags):
    """The calling thread's own choice (ngsqc_set_cram_skip_thread; -1: back to the process-wide one). Returns the previous value (-1: none)."""
    L = lib(); L.ngsqc_set_cram_skip_thread.restype = C.c_int32; L.ngsqc_set_cram_skip_thread.argtypes = [C.c_int32]
    return int(L.ngsqc_set_cram_skip_thread(int(flags)))


def cram_to_bam(cram_path, bam_path, regions=None):
    """Host only: the records of a CRAM 3.0 file as a BAM file (BGZF members with stored blocks) - what ngsqc_open hands to the device for a CRAM.
    regions [(chromosome name, start, end)]: only the slices that can hold their records (what ngsqc_open_regions decodes)."""
    class NR(C.Structure):
        _fields_ = [("chr", C.c_char_p), ("start", C.c_int32), ("end", C.c_int32)]
    L = lib(); L.ngsqc_cram_to_bam.restype = C.c_int; L.ngsqc_cram_to_bam.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_int64]
    regions = regions or []
    arr = (NR * max(len(regions), 1))(*[NR(os.fsencode(c), int(a), int(b)) for c, a, b in regions])
    rc = L.ngsqc_cram_to_bam(os.fsencode(cram_path), os.fsencode(bam_path), C.cast(arr, C.c_void_p) if regions else None, len(regions))
    if rc != 0:
        raise NgsqcError(rc, L.ngsqc_last_error(None).decode("utf-8", "replace"))


def bgzf_scan(data, threads=1):
    """BGZF member table of a BAM image (host only): structured array (file_offset, payload_offset, inflated_offset, payload_bytes, inflated_bytes, crc32)
    and the inflated size. threads > 1: the walk in pieces (falls back to the sequential walk when the pieces do not join)."""
    L = lib()
    L.ngsqc_bgzf_scan.restype = C.c_int
    L.ngsqc_bgzf_scan.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
    dt = np.dtype([("file_offset", "<u8"), ("payload_offset", "<u8"), ("inflated_offset", "<u8"), ("payload_bytes", "<u4"), ("inflated_bytes", "<u4"), ("crc32", "<u4"), ("walked_in_pieces", "<u4")])
    n, tot = C.c_int64(0), C.c_int64(0)
    rc = L.ngsqc_bgzf_scan(a.ctypes.data, a.size, int(threads), None, 0, C.byref(n), C.byref(tot))
    if rc != 0:
        raise NgsqcError(rc, L.ngsqc_last_error(None).decode("utf-8", "replace"))
    out = np.zeros(max(n.value, 1), dtype=dt)
    rc = L.ngsqc_bgzf_scan(a.ctypes.data, a.size, int(threads), out.ctypes.data, n.value, C.byref(n), C.byref(tot))
    if rc != 0:
        raise NgsqcError(rc, L.ngsqc_last_error(None).decode("utf-8", "replace"))
    return out[:n.value], int(tot.value)


def bai_assemble(bai_path, n_ref, first_record_voff, end_voff, runs, lidx, lidx_first, counts, csi_geom=None):
    """Host half of Handle.write_bai / write_csi (no device). runs: [(voff, tid, bin, pos, kind)] in file order; lidx: uint64 per window (16 kb for BAI;
    2**64 - 1: none); lidx_first: n_ref + 1 window offsets; counts: (mapped, unmapped) per reference, then of the reads without reference.
    csi_geom = (min_shift, depth): a CSI index in that geometry instead of a BAI."""
    L = lib()
    L.ngsqc_bai_assemble.restype = C.c_int
    L.ngsqc_bai_assemble.argtypes = [C.c_char_p, C.c_int32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ngsqc_csi_assemble.restype = C.c_int
    L.ngsqc_csi_assemble.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    ra = np.zeros(max(len(runs), 1), dtype=[("voff", "<u8"), ("tid", "<i4"), ("bin", "<u4"), ("pos", "<i4"), ("kind", "<u4")])
    for i, r in enumerate(runs):
        ra[i] = tuple(r)
    la = np.ascontiguousarray(lidx, dtype=np.uint64) if len(lidx) else np.zeros(1, np.uint64)
    fa = np.ascontiguousarray(lidx_first, dtype=np.int64); ca = np.ascontiguousarray(counts, dtype=np.int64).reshape(-1)
    assert fa.size == n_ref + 1 and ca.size == 2 * (n_ref + 1)
    if csi_geom is not None:
        rc = L.ngsqc_csi_assemble(os.fsencode(bai_path), int(csi_geom[0]), int(csi_geom[1]), int(n_ref), int(first_record_voff), int(end_voff), ra.ctypes.data, len(runs), la.ctypes.data,
                                  fa.ctypes.data, ca.ctypes.data)
    else:
        rc = L.ngsqc_bai_assemble(os.fsencode(bai_path), int(n_ref), int(first_record_voff), int(end_voff), ra.ctypes.data, len(runs), la.ctypes.data, fa.ctypes.data, ca.ctypes.data)
    if rc != 0:
        raise NgsqcError(rc, L.ngsqc_last_error(None).decode("utf-8", "replace"))


def plan_shard_fix(summaries, shard):
    """summaries: int64 array [n_shards, 6] (SUMMARY_FIELDS order, file order). Pure host logic of the library (no device):
    returns the ShardFix of `shard`; raises NgsqcError when the shards' record chains do not join."""
    a = np.ascontiguousarray(summaries, dtype=np.int64).reshape(-1, len(SUMMARY_FIELDS))
    arr = (ShardSummary * a.shape[0])()
    for i in range(a.shape[0]):
        for j, f in enumerate(SUMMARY_FIELDS):
            setattr(arr[i], f, int(a[i, j]))
    fix = ShardFix()
    rc = lib().ngsqc_plan_shard_fix(arr, a.shape[0], int(shard), C.byref(fix))
    if rc != 0:
        raise NgsqcError(rc, lib().ngsqc_last_error(None).decode("utf-8", "replace"))
    return fix


def _regions_array(regions):
    if isinstance(regions, C.Array):   # prepared once by the caller (bench: 200 k lines per step)
        return regions
    if isinstance(regions, np.ndarray):
        a = np.ascontiguousarray(regions, dtype=np.int32).reshape(-1, 3)
        arr = (Region * max(a.shape[0], 1)).from_buffer_copy(a.tobytes() if a.shape[0] else bytes(C.sizeof(Region)))
        return arr
    arr = (Region * max(len(regions), 1))()
    for i, (tid, s, e) in enumerate(regions):
        arr[i].tid, arr[i].start, arr[i].end = int(tid), int(s), int(e)
    return arr


class Handle:
    """One open BAM on one GPU (ngsqc_handle)."""

    def __init__(self, path=None, data=None, device=0, shard=None, voff_range=None, regions=None):
        """shard=(i, n): own the records that start inside the i-th of n contiguous BGZF-member ranges of the BAM.
        voff_range=(beg, end): the records of a virtual-offset range of the file at `path` (from bai_range: index-driven partial decode)."""
        L = lib()
        h = C.c_void_p()
        si, sn = (int(shard[0]), int(shard[1])) if shard is not None else (0, 1)
        self.shard = (si, sn)
        if regions is not None:   # [(chromosome name, start, end)] 1-based closed: header -> BAI -> one range (ngsqc_open_regions)
            class NR(C.Structure):
                _fields_ = [("chr", C.c_char_p), ("start", C.c_int32), ("end", C.c_int32)]
            arr = (NR * max(len(regions), 1))(*[NR(os.fsencode(c), int(a), int(b)) for c, a, b in regions])
            L.ngsqc_open_regions.restype = C.c_int; L.ngsqc_open_regions.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
            rc = L.ngsqc_open_regions(os.fsencode(path), device, C.cast(arr, C.c_void_p), len(regions), C.byref(h))
        elif voff_range is not None:
            L.ngsqc_open_range.restype = C.c_int; L.ngsqc_open_range.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]
            rc = L.ngsqc_open_range(os.fsencode(path), device, int(voff_range[0]), int(voff_range[1]), C.byref(h))
        elif path is not None:
            rc = L.ngsqc_open_shard(os.fsencode(path), device, si, sn, C.byref(h)) if shard is not None else L.ngsqc_open(os.fsencode(path), device, C.byref(h))
        else:
            buf = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8))
            rc = (L.ngsqc_open_memory_shard(buf.ctypes.data, buf.size, device, si, sn, C.byref(h)) if shard is not None
                  else L.ngsqc_open_memory(buf.ctypes.data, buf.size, device, C.byref(h)))
        if rc != 0:
            raise NgsqcError(rc, L.ngsqc_last_error(None).decode("utf-8", "replace"))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().ngsqc_close(self.h); self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise NgsqcError(rc, lib().ngsqc_last_error(self.h).decode("utf-8", "replace"))

    @property
    def refs(self):
        L = lib()
        return [(L.ngsqc_ref_name(self.h, i).decode(), L.ngsqc_ref_len(self.h, i)) for i in range(L.ngsqc_n_ref(self.h))]

    @property
    def n_records(self):
        n = lib().ngsqc_n_records(self.h)
        if n < 0:
            self._chk(int(n))
        return n

    @property
    def inflated_size(self): return lib().ngsqc_inflated_size(self.h)
    @property
    def n_blocks(self): return lib().ngsqc_n_bgzf_blocks(self.h)
    @property
    def compressed_size(self): return lib().ngsqc_compressed_size(self.h)

    def decode(self): self._chk(lib().ngsqc_decode(self.h))
    def drop_decoded(self): self._chk(lib().ngsqc_drop_decoded(self.h))

    def inflated(self):
        a = np.empty(self.inflated_size, dtype=np.uint8)
        self._chk(lib().ngsqc_copy_inflated(self.h, a.ctypes.data, a.size))
        return a

    def record_offsets(self):
        a = np.empty(self.n_records, dtype=np.int64)
        self._chk(lib().ngsqc_copy_record_offsets(self.h, a.ctypes.data, a.size))
        return a

    def _mapping_params(self, mode, regions=None, min_mapq=1, tid_x=-1, tid_y=-1, nonspecial=None, gc_chunks=None, gc_bin=None):
        p = MappingParams()
        p.mode, p.min_mapq, p.tid_x, p.tid_y = mode, min_mapq, tid_x, tid_y
        n_ref = len(self.refs)
        ns = np.ascontiguousarray(nonspecial if nonspecial is not None else np.zeros(n_ref, np.uint8), dtype=np.uint8)
        p.tid_nonspecial = ns.ctypes.data
        keep = [ns]
        if regions:
            ra = _regions_array(regions); keep.append(ra)
            p.regions = C.cast(ra, C.c_void_p).value; p.n_regions = len(regions)
        if gc_chunks:
            ga = _regions_array(gc_chunks); gb = np.ascontiguousarray(gc_bin, dtype=np.int32); keep += [ga, gb]
            p.gc_chunks = C.cast(ga, C.c_void_p).value; p.gc_bin = gb.ctypes.data; p.n_gc_chunks = len(gc_chunks)
        return p, keep

    def scan_mapping(self, mode, regions=None, min_mapq=1, tid_x=-1, tid_y=-1, nonspecial=None, gc_chunks=None, gc_bin=None):
        """regions / gc_chunks: lists of (tid, start, end), 1-based closed, merged+sorted. Returns (counters, gc_reads)."""
        p, keep = self._mapping_params(mode, regions, min_mapq, tid_x, tid_y, nonspecial, gc_chunks, gc_bin)
        counters = np.zeros(NCOUNTERS, dtype=np.int64)
        gc = np.zeros(101, dtype=np.float64)
        self._chk(lib().ngsqc_scan_mapping(self.h, C.byref(p), counters.ctypes.data, gc.ctypes.data))
        return counters, gc

    def run_job(self, mapping=None, depth=None, sites=None, site_params=(1, 13, False), read_qc=None, n_cycles=320, partial=False):
        """ONE pass over the BAM for every requested consumer (ngsqc_run_job; partial=True: ngsqc_run_job_partial on a shard handle - the result
        holds 'summary' (ngsqc_shard_summary as int64[6]) instead of 'counters', which come from scan_mapping_finish after the exchange).
        mapping: dict of scan_mapping keyword arguments incl. 'mode'; depth: dict(regions=, min_mapq=, min_baseq=, skip_mismapped=);
        sites: list of (tid, pos) or int32 [n, 3]; read_qc: None or dict(single_end=bool).
        Returns a dict with 'counters', 'gc_reads', 'site_counts', 'reads' for the consumers that ran."""
        jd, jr, keep, out = JobDesc(), JobResult(), [], {}
        if mapping is not None:
            kw = dict(mapping); mode = kw.pop("mode")
            p, k = self._mapping_params(mode, **kw); keep += [p, k]
            jd.mapping = C.addressof(p)
            out["counters"] = np.zeros(NCOUNTERS, dtype=np.int64); out["gc_reads"] = np.zeros(101, dtype=np.float64)
            jr.counters = out["counters"].ctypes.data; jr.gc_reads = out["gc_reads"].ctypes.data
        if depth is not None:
            dp = DepthParams(); ra = _regions_array(depth["regions"]); keep += [dp, ra]
            dp.min_mapq, dp.min_baseq, dp.skip_mismapped = depth.get("min_mapq", 1), depth.get("min_baseq", 0), int(depth.get("skip_mismapped", False))
            dp.regions = C.cast(ra, C.c_void_p).value; dp.n_regions = len(depth["regions"])
            jd.depth = C.addressof(dp)
        if sites is not None:
            if isinstance(sites, np.ndarray):
                arr = np.ascontiguousarray(sites, dtype=np.int32); n = arr.shape[0]; ptr = arr.ctypes.data
            else:
                arr = _regions_array([(t, p_, p_) for t, p_ in sites]); n = len(sites); ptr = C.cast(arr, C.c_void_p).value
            keep.append(arr)
            out["site_counts"] = np.zeros((max(n, 1), 8), dtype=np.int64)
            jd.sites = ptr; jd.n_sites = n; jd.site_min_mapq, jd.site_min_baseq, jd.site_include_npp = int(site_params[0]), int(site_params[1]), int(bool(site_params[2]))
            jr.site_counts = out["site_counts"].ctypes.data
        st = None
        if read_qc is not None:
            st = ReadStats(); jd.read_qc = 1; jd.read_qc_single_end = int(bool(read_qc.get("single_end", False))); jr.read_stats = C.addressof(st)
        if partial:
            L = lib(); L.ngsqc_run_job_partial.restype = C.c_int; L.ngsqc_run_job_partial.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            sm = ShardSummary()
            self._chk(L.ngsqc_run_job_partial(self.h, C.byref(jd), C.byref(jr), C.byref(sm)))
            out["summary"] = np.array([getattr(sm, f) for f in SUMMARY_FIELDS], dtype=np.int64); out.pop("counters", None); out.pop("gc_reads", None)
        else:
            self._chk(lib().ngsqc_run_job(self.h, C.byref(jd), C.byref(jr)))
        if sites is not None:
            out["site_counts"] = out["site_counts"][:jd.n_sites]
        if st is not None:
            out["reads"] = self._read_stats_dict(st, n_cycles)
        return out

    def depth_select(self, which):
        self._chk(lib().ngsqc_depth_select(self.h, int(which)))

    def depth_reduce(self, others):
        """this handle's difference array += the arrays of the other shard handles (peer copies, no host staging)."""
        arr = (C.c_void_p * max(len(others), 1))(*[o.h for o in others])
        self._chk(lib().ngsqc_depth_reduce(self.h, arr, len(others)))

    def _read_stats_dict(self, st, n_cycles):
        out = {f: (np.array(getattr(st, f), dtype=np.int64) if hasattr(getattr(st, f), "__len__") else int(getattr(st, f))) for f, _ in ReadStats._fields_}
        lens = np.zeros(out["max_cycles"] + 1, dtype=np.int64)
        self._chk(lib().ngsqc_read_length_hist(self.h, lens.ctypes.data, lens.size))
        cyc = np.zeros((max(n_cycles, 1), 7), dtype=np.int64)
        self._chk(lib().ngsqc_read_cycle_stats(self.h, cyc.ctypes.data, n_cycles))
        out["read_lengths"] = lens; out["cycles"] = cyc[:n_cycles]
        return out

    def scan_reads(self, single_end=False, n_cycles=320):
        """StatisticsReads::update over the whole BAM. Returns a dict of numpy arrays / ints (layout of ngsqc_read_stats) plus
        'read_lengths' (reads per length) and 'cycles' ([n, 7]: A, C, G, T, N, quality sum forward, quality sum reverse)."""
        st = ReadStats()
        self._chk(lib().ngsqc_scan_reads(self.h, int(single_end), C.byref(st)))
        return self._read_stats_dict(st, n_cycles)

    def site_pileup(self, sites, min_mapq=1, min_baseq=13, include_not_properly_paired=False):
        """sites: list of (tid, pos) sorted by tid then pos. Returns int64[n, 8]: A, C, G, T, N, deletion, other-letter, not-found."""
        # sites may also be an int32 array [n, 3] of (tid, pos, pos) rows (the C layout), e.g. prepared once for repeated calls
        if isinstance(sites, np.ndarray):
            arr = np.ascontiguousarray(sites, dtype=np.int32); n = arr.shape[0]; ptr = arr.ctypes.data
        else:
            arr = _regions_array([(t, p, p) for t, p in sites]); n = len(sites); ptr = C.cast(arr, C.c_void_p)
        out = np.zeros((max(n, 1), 8), dtype=np.int64)
        self._chk(lib().ngsqc_site_pileup(self.h, ptr, n, min_mapq, min_baseq, int(include_not_properly_paired), out.ctypes.data))
        return out[:n]

    # ---- one BAM sharded over several handles (include/ngsqc.h, "sharded" section) ----
    def scan_mapping_partial(self, mode, **kw):
        """Local scan of a shard. Returns its summary as int64[6] (SUMMARY_FIELDS order)."""
        p, keep = self._mapping_params(mode, **kw)
        sm = ShardSummary()
        self._chk(lib().ngsqc_scan_mapping_partial(self.h, C.byref(p), C.byref(sm)))
        return np.array([getattr(sm, f) for f in SUMMARY_FIELDS], dtype=np.int64)

    def scan_mapping_finish(self, fix):
        """Local fix-up after the summaries were exchanged. Returns ADDITIVE (counters, gc_reads) of this shard."""
        counters = np.zeros(NCOUNTERS, dtype=np.int64)
        gc = np.zeros(101, dtype=np.float64)
        self._chk(lib().ngsqc_scan_mapping_finish(self.h, C.byref(fix), counters.ctypes.data, gc.ctypes.data))
        return counters, gc

    def depth_device(self):
        """(device pointer, n_slots) of the un-prefixed int32 difference array (for an in-place all-reduce)."""
        ptr = C.c_void_p(); n = C.c_int64(0)
        self._chk(lib().ngsqc_depth_device(self.h, C.byref(ptr), C.byref(n)))
        return (ptr.value or 0), int(n.value)

    def depth_diff(self):
        _, n = self.depth_device()
        a = np.zeros(max(n, 1), dtype=np.int32)
        self._chk(lib().ngsqc_depth_diff_copy(self.h, a.ctypes.data, n))
        return a[:n]

    def depth_diff_set(self, a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        self._chk(lib().ngsqc_depth_diff_set(self.h, a.ctypes.data, a.size))

    def depth_finalize(self):
        self._chk(lib().ngsqc_depth_finalize(self.h))

    def scan_depth(self, regions, min_mapq=1, min_baseq=0, skip_mismapped=False, partial=False, n_regions=None):
        """partial=True: shard variant that leaves the additive difference array (see depth_diff / depth_finalize)."""
        p = DepthParams()
        ra = _regions_array(regions)
        p.min_mapq, p.min_baseq, p.skip_mismapped = min_mapq, min_baseq, int(skip_mismapped)
        p.regions = C.cast(ra, C.c_void_p).value; p.n_regions = len(regions) if n_regions is None else n_regions
        self._chk((lib().ngsqc_scan_depth_partial if partial else lib().ngsqc_scan_depth)(self.h, C.byref(p)))

    def region_read_counts(self, regions, min_mapq=1):
        """BedReadCount: reads overlapping each (merged + sorted) region."""
        ra = _regions_array(regions)
        out = np.zeros(max(len(regions), 1), dtype=np.int64)
        self._chk(lib().ngsqc_region_read_counts(self.h, C.cast(ra, C.c_void_p), len(regions), int(min_mapq), out.ctypes.data))
        return out[:len(regions)]

    def depth_stats(self, hist_cap, half_depth):
        hist = np.zeros(hist_cap + 1, dtype=np.int64)
        cov = C.c_int64(0)
        self._chk(lib().ngsqc_depth_stats(self.h, hist_cap, int(half_depth), hist.ctypes.data, C.addressof(cov)))
        return hist, cov.value

    def depth(self, roi_bases):
        out = np.zeros(max(roi_bases, 1), dtype=np.int32)
        self._chk(lib().ngsqc_depth_copy(self.h, out.ctypes.data, roi_bases))
        return out[:roi_bases]

    def region_sums(self, lines, n_lines=None):
        la = _regions_array(lines)
        n = len(lines) if n_lines is None else n_lines
        sums = np.zeros(max(n, 1), dtype=np.int64)
        self._chk(lib().ngsqc_region_sums(self.h, C.cast(la, C.c_void_p), n, sums.ctypes.data))
        return sums[:n]

    def lowhigh_runs(self, lines, cutoff, is_high=False, saturate254=False, n_lines=None, as_array=False):
        la = _regions_array(lines)
        nl = len(lines) if n_lines is None else n_lines
        n = C.c_int64(0)
        self._chk(lib().ngsqc_lowhigh_runs(self.h, C.cast(la, C.c_void_p), nl, cutoff, int(is_high), int(saturate254), None, 0, C.byref(n)))
        runs = (Run * max(n.value, 1))()
        if n.value:
            self._chk(lib().ngsqc_lowhigh_runs(self.h, C.cast(la, C.c_void_p), nl, cutoff, int(is_high), int(saturate254),
                                               C.cast(runs, C.c_void_p), n.value, C.byref(n)))
        if as_array:   # (bench: no per-run Python objects)
            return runs
        return [(runs[i].line, runs[i].start, runs[i].end) for i in range(n.value)]

    def write_bai(self, bai_path=None):
        """Writes the BAI index of the BAM (what `samtools index` writes; default <path>.bai). The handle must be on the whole file."""
        L = lib(); L.ngsqc_write_bai.restype = C.c_int; L.ngsqc_write_bai.argtypes = [C.c_void_p, C.c_char_p]
        self._chk(L.ngsqc_write_bai(self.h, os.fsencode(bai_path) if bai_path is not None else None))

    def write_csi(self, csi_path=None, min_shift=14):
        """Writes the CSI index of the BAM (what `samtools index -c -m min_shift` writes; default <path>.csi). The handle must be on the whole file."""
        L = lib(); L.ngsqc_write_csi.restype = C.c_int; L.ngsqc_write_csi.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
        self._chk(L.ngsqc_write_csi(self.h, os.fsencode(csi_path) if csi_path is not None else None, int(min_shift)))

    def upload_wait(self):
        """The compressed image is on the device (a path is copied in the background); timings()['h2d_ms'] is final behind this call."""
        L = lib(); L.ngsqc_upload_wait.restype = C.c_int; L.ngsqc_upload_wait.argtypes = [C.c_void_p]
        self._chk(L.ngsqc_upload_wait(self.h))

    def timings(self):
        t = Timings()
        self._chk(lib().ngsqc_get_timings_sized(self.h, C.byref(t), C.sizeof(t)))   # (sized: a binding older than the library gets the fields it knows)
        return {f: getattr(t, f) for f, _ in Timings._fields_}


COMM_ID_BYTES = 128


class Comm:
    """The product's own collective (include/ngsqc.h, csrc/comm.hip): RCCL over xGMI, one process per GPU. Rank 0 calls Comm.unique_id() and hands the 128 bytes to
    the other ranks (any out-of-band channel: a file, the launcher's store); every rank then makes Comm(rank, world, id, device)."""

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        rc = lib().ngsqc_comm_unique_id(C.cast(buf, C.c_void_p))
        if rc:
            raise NgsqcError(rc, lib().ngsqc_last_error(None).decode("utf-8", "replace"))
        return bytes(buf)

    def __init__(self, rank, world, uid, device=0):
        self.c = C.c_void_p(); self.rank, self.world = rank, world
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(uid)
        rc = lib().ngsqc_comm_init(rank, world, C.cast(buf, C.c_void_p), device, C.byref(self.c))
        if rc:
            raise NgsqcError(rc, lib().ngsqc_last_error(None).decode("utf-8", "replace"))

    def _chk(self, rc):
        if rc:
            raise NgsqcError(rc, lib().ngsqc_comm_last_error(self.c).decode())

    def allreduce_counters(self, counters):
        v = np.ascontiguousarray(counters, dtype=np.int64).copy()
        assert v.size == NCOUNTERS
        self._chk(lib().ngsqc_comm_allreduce_counters(self.c, v.ctypes.data)); return v

    def allreduce_i64(self, v, take_max=False):
        a = np.ascontiguousarray(v, dtype=np.int64).copy()
        self._chk(lib().ngsqc_comm_allreduce_i64(self.c, a.ctypes.data, a.size, int(take_max))); return a

    def allreduce_f64(self, v):
        a = np.ascontiguousarray(v, dtype=np.float64).copy()
        self._chk(lib().ngsqc_comm_allreduce_f64(self.c, a.ctypes.data, a.size)); return a

    def allgather_summaries(self, mine):
        """mine: int64[6] (ngsqc_shard_summary) -> int64[world, 6] in rank order"""
        m = ShardSummary(*[int(x) for x in mine]); allv = (ShardSummary * self.world)()
        self._chk(lib().ngsqc_comm_allgather_summaries(self.c, C.byref(m), allv))
        return np.array([[getattr(x, f) for f in SUMMARY_FIELDS] for x in allv], dtype=np.int64)

    def allreduce_depth(self, handle):
        self._chk(lib().ngsqc_comm_allreduce_depth(self.c, handle.h))

    def close(self):
        if self.c:
            lib().ngsqc_comm_destroy(self.c); self.c = C.c_void_p()
