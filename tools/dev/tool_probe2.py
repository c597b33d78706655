"""dev: bin/MappingQC -wgs on a generated BAM in /dev/shm under several environments (one line per run: wall, open, job, K1, close).
usage: tool_probe2.py reads "K=V,K=V" "K=V" ... (an empty string = defaults)"""
import os, re, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import bamgen_lib as G
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96_000_000
f = f"/dev/shm/ngsqc_tool_{n}.bam"
G.generate(n).tofile(f); open(f + ".bai", "wb").close()
try:
    for spec in sys.argv[2:] or [""]:
        env = dict(os.environ, NGSQC_TIMING="1")
        for kv in filter(None, spec.split(",")):
            k, v = kv.split("=", 1); env[k] = v
        t = time.perf_counter()
        p = subprocess.run([os.path.join(R, "ngs-bits_amd", "bin", "MappingQC"), "-in", f, "-wgs", "-build", "hg38", "-out", "/tmp/tool_probe.qcML", "-no_ref"], env=env, capture_output=True, text=True)
        w = time.perf_counter() - t
        st = {m.group(2): float(m.group(1)) for m in re.finditer(r"\+([0-9.]+) s ([a-z ]+: [a-z]+)", p.stderr)}
        k1 = re.search(r"fused job: ([0-9.]+) ms wall, K1 ([0-9.]+) ms, (\d+) tiles", p.stderr)
        print(f"[tool] {spec or 'defaults':60s} rc {p.returncode} wall {w:.3f} s  open {st.get('open: done', -1):.3f}  job {st.get('fused job: done', 0) - st.get('fused job: start', 0):.3f}  "
              f"close {st.get('close: done', 0) - st.get('close: start', 0):.3f}  K1 {k1.group(2) if k1 else '?'} ms  tiles {k1.group(3) if k1 else '?'}", flush=True)
finally:
    for x in (f, f + ".bai", "/tmp/tool_probe.qcML"):
        if os.path.exists(x): os.remove(x)
