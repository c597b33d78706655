"""dev: where the wall time of bin/MappingQC -wgs goes on a generated BAM in /dev/shm (NGSQC_TIMING stamps + NGSQC_DEBUG tile lines of the library)"""
import os, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import bamgen_lib as G
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96_000_000
f = f"/dev/shm/ngsqc_tool_{n}.bam"
G.generate(n).tofile(f); open(f + ".bai", "wb").close()
try:
    for i in range(2):
        env = dict(os.environ, NGSQC_TIMING="1", NGSQC_DEBUG="1")
        if i == 1 and len(sys.argv) > 2:
            for kv in sys.argv[2:]:
                k, v = kv.split("=", 1); env[k] = v
        t = time.perf_counter()
        p = subprocess.run([os.path.join(R, "ngs-bits_amd", "bin", "MappingQC"), "-in", f, "-wgs", "-build", "hg38", "-out", "/tmp/tool_probe.qcML", "-no_ref"], env=env, capture_output=True, text=True)
        w = time.perf_counter() - t
        lines = [ln for ln in p.stderr.splitlines() if "[ngsqc]" in ln]
        keep = [ln for ln in lines if "tile " not in ln] + [ln for ln in lines if "tile " in ln][:6]
        print(f"== run {i}: wall {w:.3f} s, rc {p.returncode}")
        for ln in keep: print("  ", ln[:230])
finally:
    for x in (f, f + ".bai", "/tmp/tool_probe.qcML"):
        if os.path.exists(x): os.remove(x)
