"""Host decode rate of the CRAM container layer (csrc/cram.hip through ngsqc_cram_to_bam; no GPU): the reference's CRAM fixtures, slices on 1 thread and on the default pool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ngsqc = __import__("importlib").import_module("ngs-bits_amd")
GI = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "ref_in")
os.environ["NGSQC_CRAM_NO_REFERENCE"] = "1"
print("# file\tcram_bytes\tbam_stream_bytes\tslices\tthreads\tbest_of_5_ms\tMB_of_records_per_s")
for name, slices in (("cramTest.cram", 7), ("SampleIdentity_in_wes.cram", 5), ("SampleIdentity_in_rna.cram", 4)):
    src = os.path.join(GI, name)
    for th in ("1", ""):
        if th: os.environ["NGSQC_CRAM_THREADS"] = th
        else: os.environ.pop("NGSQC_CRAM_THREADS", None)
        best = 1e9
        for _ in range(5):
            t = time.perf_counter(); ngsqc.cram_to_bam(src, "/tmp/_rate.bam"); best = min(best, time.perf_counter() - t)
        out = os.path.getsize("/tmp/_rate.bam")
        print("%s\t%d\t%d\t%d\t%s\t%.1f\t%.0f" % (name, os.path.getsize(src), out, slices, th or "pool", best * 1e3, out / best / 1e6))
