import glob, os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ngsqc = importlib.import_module("ngs-bits_amd")
bams = sorted(p[:-4] for p in glob.glob("tests/golden/ref_in/*.bam.bai"))
for bam in bams:
    sys.stderr.write("== " + bam + "\n"); sys.stderr.flush()
    h = ngsqc.Handle(path=bam)
    try:
        h.write_bai("/tmp/dbg.bai")
    finally:
        h.close()
    sys.stderr.write("   written %d\n" % os.path.getsize("/tmp/dbg.bai")); sys.stderr.flush()
