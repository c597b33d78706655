#!/usr/bin/env python3
"""Dev probe: first-touch (page fault) bandwidth of the host with T threads, and BAM generator scaling."""
import ctypes, mmap, os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
libc = ctypes.CDLL("libc.so.6")
libc.memset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]; libc.memset.restype = ctypes.c_void_p
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print("[cgroup]", f, open(f).read().strip(), flush=True)
    except OSError:
        pass
print("[affinity]", len(os.sched_getaffinity(0)), flush=True)
import bamgen_lib as G
for T in (16, 32, 48, 64, 96):
    t0 = time.time(); a = G.generate(8_000_000, threads=T); print(f"[gen] 8M reads, {T} threads: {time.time() - t0:.1f} s", flush=True); del a
