"""probe: ngsqc_open_head on a BAM whose header spans several BGZF members and whose first record starts inside a member (tests/cram_twin.py writes such files)"""
import ctypes as C, os, struct, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
ngsqc = __import__("importlib").import_module("ngs-bits_amd")
import cram_twin
t = cram_twin.make_twin(os.path.join(ROOT, "tests/golden/ref_in/MappingQC_in2.bam"), "/tmp/head_probe", max_records=20000)
L = ngsqc.capi.lib()
for head in (8, 32):
    h = C.c_void_p()
    L.ngsqc_open_head.restype = C.c_int; L.ngsqc_open_head.argtypes = [C.c_char_p, C.c_int, C.c_int64, C.POINTER(C.c_void_p)]
    rc = L.ngsqc_open_head(t["bam"].encode(), 0, head, C.byref(h)); print("open_head", head, "rc", rc, L.ngsqc_last_error(None))
    L.ngsqc_n_records.restype = C.c_int64; L.ngsqc_n_records.argtypes = [C.c_void_p]; L.ngsqc_inflated_size.restype = C.c_int64; L.ngsqc_inflated_size.argtypes = [C.c_void_p]
    n = L.ngsqc_n_records(h); nb = L.ngsqc_inflated_size(h); print(" n_records", n, "inflated", nb, "last_error", L.ngsqc_last_error(h))
    if n > 0:
        infl = np.zeros(nb, np.uint8); off = np.zeros(n, np.int64)
        L.ngsqc_copy_inflated.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]; L.ngsqc_copy_record_offsets.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        print(" copy", L.ngsqc_copy_inflated(h, infl.ctypes.data, nb), L.ngsqc_copy_record_offsets(h, off.ctypes.data, n))
        print(" offsets", off[:4], "first record bytes", bytes(infl[off[0]:off[0] + 40]).hex(), "expected", t["records"][0][:40].hex())
    L.ngsqc_close.argtypes = [C.c_void_p]; L.ngsqc_close(h)
p = subprocess.run([os.path.join(ROOT, "ngs-bits_amd/bin/BamInfo"), "-in", t["bam"], "-name"], capture_output=True, text=True, env=dict(os.environ, NGSQC_TIMING="1", NGSQC_DEBUG="1"))
print(p.stdout); print(p.stderr[-3000:])
