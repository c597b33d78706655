"""BASELINE.json configs[0] on the device: the same 100k-read chr21 exome-subset BAM as tests/test_cpu_config0.py through the MappingQC binary (every line of its
TXT output) and through the C ABI (counters, per-base depth), against the oracle. Statistics::mapping, src/cppNGS/Statistics.cpp:346-700; MappingQC main,
src/MappingQC/main.cpp."""
import os
import subprocess

import numpy as np
import pytest

import config0_chr21 as C0
import hostprep as H
import oracle_lib as O
from conftest import ROOT

pytestmark = pytest.mark.gpu
ngsqc = __import__("importlib").import_module("ngs-bits_amd")
BIN = os.path.join(ROOT, "ngs-bits_amd", "bin")


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    return C0.write_inputs(tmp_path_factory.mktemp("config0"))


def test_mappingqc_tool_output(inputs, tmp_path):
    bam, bed = inputs
    exp = O.mapping(O.Bam(bam), O.MODE_ROI, bed, merge_bed=True).values()
    for shards in (None, "4"):
        out = str(tmp_path / f"qc_{shards}.txt")
        env = dict(os.environ, **({"NGSQC_SHARDS": shards} if shards else {}))
        p = subprocess.run([os.path.join(BIN, "MappingQC"), "-in", bam, "-roi", bed, "-no_ref", "-no_cont", "-txt", "-out", out], capture_output=True, text=True, timeout=300, env=env)
        assert p.returncode == 0, p.stderr
        got = dict(ln.split(": ", 1) for ln in open(out).read().splitlines() if ": " in ln)
        dropout = {"AT dropout", "GC dropout"}   # (-no_ref: "n/a (no reference genome)" instead of a value)
        assert {k: v for k, v in got.items() if k not in dropout} == {k: v for k, v in exp.items() if k not in dropout}, sorted(set(got.items()) ^ set(exp.items()))
        assert all(got[k].startswith("n/a") for k in dropout)


def test_counters_and_depth_through_the_c_abi(inputs):
    bam, bed = inputs
    ob = O.Bam(bam)
    exp = O.mapping(ob, O.MODE_ROI, bed, merge_bed=True)
    h = ngsqc.Handle(path=bam)
    try:
        assert h.n_records == C0.N_READS
        regs, _ = H.bed_regions(bed, h.refs, 1)
        tx, ty = H.xy_tids(h.refs)
        counters, _ = h.scan_mapping(ngsqc.MODE_ROI, regions=regs, tid_x=tx, tid_y=ty, nonspecial=H.nonspecial(h.refs))
        skip = {O.COUNTER_NAMES.index("half_depth"), O.COUNTER_NAMES.index("bases_covered_half")}   # (derived on the host from the depth array)
        assert [int(c) for i, c in enumerate(counters) if i not in skip] == [int(c) for i, c in enumerate(exp.counters[:len(counters)]) if i not in skip]
        assert np.array_equal(h.depth(int(exp["roi_bases"])), exp.depth)
        got, depth = C0.restate(ob.inflated(), ob.record_offsets(), bed)     # and the second witness directly
        assert np.array_equal(h.depth(int(exp["roi_bases"]))[600:-300], depth)
    finally:
        h.close()
