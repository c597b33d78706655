"""ctypes binding of tools/libbamgen.so — synthetic BAM images (SURVEY.md §8(d) shapes) for tests and bench.py."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "tools")
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(TOOLS, "libbamgen.so")
        src = os.path.join(TOOLS, "bamgen.cpp")
        if not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
            subprocess.check_call(["make", "-C", TOOLS, "-s"])
        L = C.CDLL(so)
        L.bamgen_generate_map.restype = C.c_void_p
        L.bamgen_generate_map.argtypes = [C.c_int64, C.c_uint64, C.c_int, C.c_double, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.bamgen_generate_map2.restype = C.c_void_p
        L.bamgen_generate_map2.argtypes = [C.c_int64, C.c_uint64, C.c_int, C.c_double, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.bamgen_release.argtypes = [C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


def effective_cpus():
    """CPUs this process may actually use: the cgroup CPU quota when there is one (a 256-thread host behind a 16-CPU quota runs 16), else
    the affinity mask / os.cpu_count()."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(round(int(q) / int(p)))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(round(q / p))))
        except (OSError, ValueError):
            pass
    return n


class _Mapping:
    """Owner of the generator's anonymous mapping: the numpy view keeps it alive through its base chain."""

    def __init__(self, ptr, n, cap):
        self.ptr, self.n, self.cap = ptr, n, cap
        self.__array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 3}

    def __del__(self):
        if self.ptr:
            lib().bamgen_release(self.ptr, self.cap)
            self.ptr = None


def generate(n_reads, seed=20260821, mode=0, depth=30.0, first_contig=0, start_pos=0, level=6, aligned=True, threads=0, flavor=0):
    """Returns the BAM file image as a numpy uint8 array (a zero-copy view of the generator's buffer).
    mode 0 = short-read WGS, 1 = ONT-like long reads. flavor (short reads): bit 0 = SEQ from a synthetic reference genome (overlapping reads share
    sequence), bits 1-2 = quality model (0: four levels as in SURVEY.md 8(d), 1: eight bins, 2: forty levels)."""
    if threads <= 0:
        threads = 2 * effective_cpus()   # (oversubscribing a CPU quota costs: 256 threads on a 16-CPU quota were 1.5x slower than 32)
    n, cap = C.c_size_t(0), C.c_size_t(0)
    p = lib().bamgen_generate_map2(n_reads, seed, mode, depth, first_contig, start_pos, level, int(aligned), threads, int(flavor), C.byref(n), C.byref(cap))
    if not p:
        raise MemoryError(f"bamgen: could not map {cap.value} bytes for {n_reads} reads")
    return np.asarray(_Mapping(p, n.value, cap.value))


def write(path, **kw):
    a = generate(**kw)
    a.tofile(path)
    return a.size
