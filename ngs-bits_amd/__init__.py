"""ngs-bits_amd — MI355X-native BAM mapping-QC / coverage hot path (drop-in for one path of imgag/ngs-bits).

The product is the C-ABI library libngsqc_hip.so (include/ngsqc.h) built from csrc/*.hip plus the C++ host tools under
host/. This Python package is only a thin ctypes binding used by tests/ and bench.py; it fails loudly if the HIP
library has not been built (there is no CPU fallback).
"""
from .capi import (  # noqa: F401
    NgsqcError, Handle, Region, MappingParams, lib, lib_path, build_library,
    MODE_ROI, MODE_NOROI, MODE_WGS, NCOUNTERS, COUNTER_NAMES, ShardSummary, ShardFix, SUMMARY_FIELDS, plan_shard_fix, bai_range, bai_ranges, set_reference, set_cram_skip, set_cram_skip_thread, CRAM_SKIP_NAMES, CRAM_SKIP_TAGS, cram_to_bam, bai_assemble, bgzf_scan, Comm, device_count,
)
from .dist import allreduce_counters, combine_counters_local, shard_blocks, scan_mapping_sharded, scan_mapping_sharded_local, scan_depth_sharded_local  # noqa: F401,E402
