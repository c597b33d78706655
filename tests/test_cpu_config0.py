"""BASELINE.json configs[0]: MappingQC on a 100k-read chr21 exome-subset BAM through the CPU reference path (plumbing, no GPU). The CPU path here is the oracle
(oracle/stats.hpp mapping_roi, the restatement of Statistics::mapping, src/cppNGS/Statistics.cpp:346-700); it is held against a second, array-wise restatement of
the same reference lines (tests/config0_chr21.py) so that this instance has a witness that is not the oracle itself."""
import numpy as np
import pytest

import config0_chr21 as C0
import oracle_lib as O


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    return C0.write_inputs(tmp_path_factory.mktemp("config0"))


def test_oracle_against_the_array_restatement(inputs):
    bam, bed = inputs
    ob = O.Bam(bam)
    assert ob.count == C0.N_READS
    assert ob.refs[C0.CHR21][0] == "chr21"
    exp = O.mapping(ob, O.MODE_ROI, bed, merge_bed=True)
    got, depth = C0.restate(ob.inflated(), ob.record_offsets(), bed)
    for k, v in got.items():
        assert exp[k] == v, (k, exp[k], v)
    roi = C0.merged_chr21(bed)
    assert exp["roi_bases"] == int((roi[:, 1] - roi[:, 0] + 1).sum()) + 600 + 300          # + the two targets without reads (chr1 before chr21, chrX after it)
    assert not exp.depth[:600].any() and not exp.depth[-300:].any()
    assert np.array_equal(exp.depth[600:-300], depth)
    # the shape the config names: reads over the whole chromosome, an exome-like target of well under 1 % of it
    assert 0.003 < got["al_ontarget"] / got["al_total"] < 0.05 and 0.1 < depth.mean() < 0.5 and len(roi) > 1900
    v = exp.values()
    assert v["on-target read percentage"] == "%.2f" % (100.0 * got["al_ontarget"] / got["al_total"])
    assert v["target region read depth"] == "%.2f" % (got["bases_usable"] / exp["roi_bases"])
