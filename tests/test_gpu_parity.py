"""GPU parity: libngsqc_hip.so (through the C ABI) vs the CPU oracle, bit-exact, on the reference's own fixture BAMs."""
import os

import numpy as np
import pytest

import oracle_lib as O
import hostprep as H
from conftest import GOLDEN_IN as GI, RESOURCES

pytestmark = pytest.mark.gpu

ngsqc = __import__("importlib").import_module("ngs-bits_amd")

BAMS = ["close_exons.bam", "MappingQC_in1.bam", "MappingQC_in2.bam", "MappingQC_in3.bam", "MappingQC_in4.bam", "MappingQC_in5.bam",
        "Statistics_mapqc_wgs.bam", "Statistics_longread.bam", "BamReader_lr.bam", "BamReader_rna.bam", "BamReader_insert_only.bam",
        "BamReader_sr.bam", "lowcov_bug_case1.bam", "lowcov_bug_case2.bam", "sry.bam"]

SKIP_COUNTERS = {"half_depth", "bases_covered_half"}  # come from ngsqc_depth_stats


def p(name):
    return os.path.join(GI, name)


@pytest.mark.parametrize("bam", BAMS)
def test_inflate_and_record_index(bam):
    ob = O.Bam(p(bam))
    h = ngsqc.Handle(path=p(bam))
    assert [r for r in h.refs] == ob.refs
    assert h.inflated_size == ob.inflated_size
    assert np.array_equal(h.inflated(), ob.inflated())
    assert h.n_records == ob.count
    assert np.array_equal(h.record_offsets(), ob.record_offsets())
    h.close()


_FASTA = {}


def _fasta(bed, refs, tmp_root):
    """one synthetic genome per (BED, reference set): bases under every BED line (tools/fastagen.py), contigs = the BAM's references"""
    key = (bed, tuple(refs))
    if key not in _FASTA:
        path = os.path.join(tmp_root, f"genome_{len(_FASTA)}.fa")
        _FASTA[key] = H.sparse_fasta_for(bed, refs, path, seed=11 + len(_FASTA))
    return _FASTA[key]


def _compare_mapping(bam, mode, bed, merge_mode, cfdna=False, min_mapq=1, tmp_root=None):
    ob = O.Bam(p(bam))
    h = ngsqc.Handle(path=p(bam))
    refs = h.refs
    regs = gc = bins = fasta = None
    if bed:
        regs, _ = H.bed_regions(bed, refs, merge_mode)
        # real GC bins of roi.chunk(100) from a synthetic genome: the (bin, n) hit table, the n >= 64 path and the host-side reconstruction
        # of gc_reads all take part (Statistics.cpp:363-387, 533-541, 1164-1171)
        # (the OMIM regions are 416 k chunks: the oracle's FASTA leg takes ~40 s there, so only one OMIM case runs with a genome)
        lines = [ln.split("\t") for ln in open(bed) if ln.strip() and not ln.startswith("#")]
        small = sum(int(f[2]) - int(f[1]) for f in lines) < 2_000_000
        lens = {H.chr_num(n): l for n, l in refs}
        inside = all(int(f[2]) <= lens.get(H.chr_num(f[0]), 1 << 40) for f in lines)   # (a BED line behind a contig end makes FastaFileIndex::seq throw, in the reference too)
        if inside and (small or bam == "MappingQC_in5.bam"):
            fasta = _fasta(bed, refs, tmp_root)
            gc, bins = H.gc_inputs(bed, refs, fasta, merge_mode)
            assert any(b >= 0 for b in bins)
    tx, ty = H.xy_tids(refs)
    counters, gc_reads = h.scan_mapping(mode, regions=regs, min_mapq=min_mapq, tid_x=tx, tid_y=ty, nonspecial=H.nonspecial(refs),
                                        gc_chunks=gc, gc_bin=bins)
    exp = O.mapping(ob, mode, bed, merge_bed=(merge_mode == 1), fasta=fasta, min_mapq=min_mapq, cfdna=cfdna)
    if fasta:
        want = np.zeros(101); want[:exp.gc_reads.size] = exp.gc_reads
        assert exp.have_gc and np.allclose(gc_reads, want, rtol=1e-12, atol=0.0), (bam, float(np.abs(gc_reads - want).max()))
        assert want.sum() > 0 or counters[O.COUNTER_NAMES.index("al_ontarget")] == 0
    for i, name in enumerate(O.COUNTER_NAMES):
        if name in SKIP_COUNTERS:
            continue
        assert int(counters[i]) == int(exp.counters[i]), (bam, name, int(counters[i]), int(exp.counters[i]))
    assert np.array_equal(counters[32:], exp.counters[32:]), "insert-size histogram"
    if regs:
        roi_bases = int(counters[O.COUNTER_NAMES.index("roi_bases")])
        d = h.depth(roi_bases)
        assert np.array_equal(d, exp.depth), "per-base depth"
        half = exp["half_depth"]
        hist, cov = h.depth_stats(2499, half)
        assert np.array_equal(hist, np.bincount(np.minimum(exp.depth, 2499), minlength=2500))
        assert cov == exp["bases_covered_half"]
    h.close()


@pytest.mark.parametrize("bam,bed,cfdna", [
    ("close_exons.bam", "close_exons.bed", False), ("MappingQC_in2.bam", "MappingQC_in2.bed", False),
    ("MappingQC_in1.bam", "MappingQC_in2.bed", False), ("MappingQC_in4.bam", "MappingQC_in3.bed", True),
    ("MappingQC_in3.bam", "MappingQC_in2.bed", False), ("Statistics_longread.bam", "panel.bed", False),
])
def test_mapping_roi(bam, bed, cfdna, tmp_path_factory):
    _compare_mapping(bam, ngsqc.MODE_ROI, p(bed), 1, cfdna, tmp_root=str(tmp_path_factory.getbasetemp()))


@pytest.mark.parametrize("bam", ["close_exons.bam", "MappingQC_in3.bam", "MappingQC_in1.bam", "Statistics_longread.bam", "BamReader_lr.bam"])
def test_mapping_noroi(bam):
    _compare_mapping(bam, ngsqc.MODE_NOROI, None, 0)


@pytest.mark.parametrize("bam,bed", [
    ("Statistics_mapqc_wgs.bam", p("Statistics_mapqc_wgs.bed")), ("close_exons.bam", None),
    ("MappingQC_in5.bam", os.path.join(RESOURCES, "hg38_440_omim_genes.bed")),
    ("MappingQC_in2.bam", os.path.join(RESOURCES, "hg19_439_omim_genes.bed")),
    ("Statistics_longread.bam", os.path.join(RESOURCES, "hg38_440_omim_genes.bed")),
])
def test_mapping_wgs(bam, bed, tmp_path_factory):
    _compare_mapping(bam, ngsqc.MODE_WGS, bed, 3 if bed else 0, tmp_root=str(tmp_path_factory.getbasetemp()))


@pytest.mark.parametrize("bam,bed,mapq,baseq", [
    ("close_exons.bam", "close_exons.bed", 1, 0), ("close_exons.bam", "close_exons.bed", 20, 20),
    ("MappingQC_in2.bam", "MappingQC_in2.bed", 1, 0), ("MappingQC_in2.bam", "MappingQC_in2.bed", 20, 30),
    ("MappingQC_in4.bam", "MappingQC_in3.bed", 1, 25), ("Statistics_longread.bam", "panel.bed", 1, 10),
])
def test_depth_tools(bam, bed, mapq, baseq):
    """BedLowCoverage/BedHighCoverage (random access and sweep) + BedCoverage cores vs oracle."""
    ob = O.Bam(p(bam))
    h = ngsqc.Handle(path=p(bam))
    regs, annos = H.bed_regions(p(bed), h.refs, 2)
    regs_ok = [r for r in regs if r[0] >= 0]
    if len(regs_ok) != len(regs):
        pytest.skip("BED chromosome missing in BAM (the reference throws)")
    h.scan_depth(regs, min_mapq=mapq, min_baseq=baseq)
    for is_high in (False, True):
        for ra in (True, False):
            exp = O.low_high_coverage(ob, p(bed), 20, mapq, baseq, is_high=is_high, random_access=ra, tool_merge=1)
            runs = h.lowhigh_runs(regs, 20, is_high=is_high, saturate254=not ra)
            # emulate the final merge(true,true,true) on the raw runs: adjacent runs of different lines with a gap of 0 merge
            got = [(regs[l][0], s, e) for (l, s, e) in runs]
            exp_runs = []
            tm = {H.chr_num(n): i for i, (n, _) in reversed(list(enumerate(h.refs)))}
            for ln in exp["bed"].splitlines():
                f = ln.split("\t"); exp_runs.append((tm[H.chr_num(f[0])], int(f[1]) + 1, int(f[2])))
            merged = []
            for r in got:
                if merged and merged[-1][0] == r[0] and merged[-1][2] + 1 >= r[1]:
                    merged[-1] = (r[0], merged[-1][1], max(merged[-1][2], r[2]))
                else:
                    merged.append(r)
            assert merged == exp_runs, (is_high, ra)
            d = h.depth(exp["roi_bases"])
            ed = exp["depth"] if ra else np.minimum(exp["depth"], 254)
            assert np.array_equal(np.minimum(d, 254) if not ra else d, ed)
    if baseq == 0:
        cov, _, _ = O.avg_coverage(ob, p(bed), merge_bed=False, min_mapq=mapq, random_access=True)
        lines, _ = H.bed_regions(p(bed), h.refs, 0)
        assert np.array_equal(h.region_sums(lines), cov)
    h.close()


@pytest.mark.parametrize("bam,bed,mapq", [("close_exons.bam", "close_exons.bed", 1), ("MappingQC_in4.bam", "MappingQC_in3.bed", 30), ("MappingQC_in2.bam", "MappingQC_in2.bed", 0)])
def test_region_read_counts(bam, bed, mapq):
    """BedReadCount core through the C ABI vs the oracle (src/BedReadCount/main.cpp:33-71)."""
    h = ngsqc.Handle(path=p(bam))
    exp, text = O.read_counts(O.Bam(p(bam)), p(bed), mapq)
    # the oracle merged the BED (merge(false)): take its lines as the regions
    tm = H.tid_map(h.refs)
    regs = [(tm.get(H.chr_num(f[0]), -1), int(f[1]) + 1, int(f[2])) for f in (ln.split("\t") for ln in text.splitlines())]
    keep = [i for i, r in enumerate(regs) if r[0] >= 0]
    got = h.region_read_counts([regs[i] for i in keep], mapq)
    assert np.array_equal(got, exp[keep]) and all(exp[i] == 0 for i in range(len(regs)) if i not in keep)
    assert int(exp.sum()) > 0
    h.close()
