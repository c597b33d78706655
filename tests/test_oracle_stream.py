"""The streaming cpu_baseline loop (oracle/stream.hpp) must give the same counters as the pinned oracle (stats.hpp)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import GOLDEN_IN as GI, RESOURCES


@pytest.mark.parametrize("bam,bed", [
    ("Statistics_mapqc_wgs.bam", os.path.join(GI, "Statistics_mapqc_wgs.bed")),
    ("MappingQC_in5.bam", os.path.join(RESOURCES, "hg38_440_omim_genes.bed")),
    ("MappingQC_in1.bam", None), ("Statistics_longread.bam", None),
])
def test_stream_equals_oracle(bam, bed):
    path = os.path.join(GI, bam)
    exp = O.mapping(O.Bam(path), O.MODE_WGS, bed, merge_bed=False)
    img = np.fromfile(path, dtype=np.uint8)
    got, st, secs = O.baseline_wgs_stream(img, bed)
    keep = np.ones(got.size, dtype=bool)
    if bed is None:
        keep[27:29] = False  # half depth is undefined without a ROI (0/0 in the reference)
    assert np.array_equal(got[keep], exp.counters[keep])
    assert st["n_records"] == O.Bam(path).count and secs >= 0


def test_all_cores_baseline_visits_every_record_once():
    """bench.py's cpu_baseline_all_cores: member ranges per thread must cover every record exactly once."""
    import bamgen_lib as G
    img = G.generate(150_000, seed=8, start_pos=15_900_000)
    bed = os.path.join(RESOURCES, "hg38_440_omim_genes.bed")
    c1, st1, _ = O.baseline_wgs_stream(img, bed, 1, -1)
    additive = np.ones(c1.size, dtype=bool); additive[list(O.ORDER_DEPENDENT)] = False
    for threads in (1, 3, 8):
        st, secs = O.baseline_wgs_stream_mt(img, bed, 1, threads)
        assert st["n_records"] == st1["n_records"] == 150_000 and st["inflated"] == st1["inflated"] and secs > 0, (threads, st, st1)
        # the additive counters of the per-thread member ranges sum to the sequential loop's (what bench.py checks the GPU against at full size)
        st, secs, cs, hist = O.baseline_wgs_stream_mt(img, bed, 1, threads, want_counters=True)
        assert np.array_equal(cs[additive], c1[additive]), threads
        assert int(hist.sum()) == int(c1[26]) and int((hist * np.arange(hist.size)).sum()) > 0


@pytest.mark.parametrize("mode,include_npp", [(0, False), (0, True), (1, True)])
def test_stream_with_the_contamination_pileup_riding(tmp_path, mode, include_npp):
    """bench.py's cpu_baseline runs the job the GPU step runs: the known-site pileup of MappingQC's third pass (Statistics::contamination -> BamReader::getPileup) rides the
    streaming loop. Its counts must be the indexed oracle's (site_pileup of stats.hpp, pinned by BamReader_Test.cpp:256-292), the mapping counters must not change."""
    import bamgen_lib as G
    import hostprep as H
    p = str(tmp_path / "s.bam")
    G.write(p, **(dict(n_reads=60_000, seed=41, start_pos=15_900_000, flavor=1) if mode == 0 else dict(n_reads=600, seed=42, mode=1, depth=40.0, start_pos=15_900_000)))
    ob = O.Bam(p)
    known = H.known_sites(ob.refs)
    lo, hi = 15_900_000, 15_900_000 + (60_000 * 150 // 30 if mode == 0 else 400_000)
    near = known[(known[:, 0] == 0) & (known[:, 1] >= lo - 1000) & (known[:, 1] <= hi + 1000)]
    extra = np.array([[0, q, 0] for q in range(lo + 500, hi, 997)], dtype=known.dtype).reshape(-1, known.shape[1])   # (the known sites are sparse: a regular grid of positions as well)
    sites = np.concatenate([near, extra, known[known[:, 0] == 5][:50]])
    bed = os.path.join(RESOURCES, "hg38_440_omim_genes.bed")
    img = np.fromfile(p, dtype=np.uint8)
    c0, _, _ = O.baseline_wgs_stream(img, bed, 1, -1)
    c1, st, secs = O.baseline_wgs_stream(img, bed, 1, -1, sites=sites, site_params=(1, 13, include_npp))
    assert np.array_equal(c0, c1) and secs > 0
    want = O.site_pileup(ob, [(int(t), int(q)) for t, q in sites[:, :2]], 1, 13, include_npp)
    assert np.array_equal(st["site_counts"], want) and int(want.sum()) > 1000
