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
