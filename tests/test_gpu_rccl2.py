"""RCCL with more than one rank (include/ngsqc.h ngsqc_comm_*): a test that ARMS ITSELF - it skips on a box with one GPU (every box of the rounds
so far) and, on the first lease with two, runs the library's communicator with one process per device: every collective of the two protocols against numpy, and the
one-BAM-over-two-GPUs MappingQC job (mapping scan + contamination pileup through the shard protocol, the int32 difference array all-reduced in place on the devices)
against the unsharded job of one GPU. RCCL refuses two ranks on one device, so nothing less than two GPUs exercises csrc/comm.hip with N > 1."""
import os
import subprocess
import sys

import numpy as np
import pytest

import importlib

ngsqc = importlib.import_module("ngs-bits_amd")
import bamgen_lib as G
import hostprep as H
from conftest import ROOT

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
OMIM = os.path.join(ROOT, "ngs-bits_amd", "resources", "hg38_440_omim_genes.bed")
SKIP = {ngsqc.COUNTER_NAMES.index("half_depth"), ngsqc.COUNTER_NAMES.index("bases_covered_half")}


def _device_count():
    # the library's own count: importing torch HERE loads torch's bundled HIP / HSA runtime beside /opt/rocm's, and the RCCL that libngsqc_hip.so opens later in this
    # process (test_gpu_shard.py) then finds an HSA that was never initialised ("pfn_hsa_system_get_info failed with 4107", seen on the GPU box in round 6)
    return ngsqc.device_count()


def test_device_count_sees_the_box():
    """include/ngsqc.h ngsqc_device_count: at least the device the GPU suite runs on, and a handle opens on the last one it names"""
    n = _device_count()
    assert n >= 1
    h = ngsqc.Handle(path=os.path.join(ROOT, "tests", "golden", "ref_in", "Statistics_mapqc_wgs.bam"), device=n - 1)
    try:
        assert h.n_records > 0
    finally:
        h.close()
    with pytest.raises(ngsqc.NgsqcError):
        ngsqc.Handle(path=os.path.join(ROOT, "tests", "golden", "ref_in", "Statistics_mapqc_wgs.bam"), device=n)


def test_two_ranks_over_rccl(tmp_path):
    world = 2
    if _device_count() < world:
        pytest.skip("one GPU on this box: RCCL refuses two ranks on one device (the test runs on the first lease with two)")
    bam = str(tmp_path / "two.bam")
    G.write(bam, n_reads=400_000, seed=23, start_pos=15_900_000)
    # the unsharded job on device 0
    h = ngsqc.Handle(path=bam, device=0)
    regs, _ = H.bed_regions(OMIM, h.refs, 3); tx, ty = H.xy_tids(h.refs)
    mp = dict(mode=ngsqc.MODE_WGS, regions=regs, min_mapq=1, tid_x=tx, tid_y=ty, nonspecial=H.nonspecial(h.refs))
    ref = h.run_job(mapping=mp, sites=H.known_sites(h.refs))
    d_ref = h.depth(int(ref["counters"][26])).copy()
    h.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    uid_file = str(tmp_path / "uid.bin")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "rccl_worker.py"), str(r), str(world), bam, uid_file, str(tmp_path / f"r{r}.npz")],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    res = [np.load(str(tmp_path / f"r{r}.npz")) for r in range(world)]
    v = np.arange(64, dtype=np.int64) * 7 - 100
    for r in range(world):
        assert np.array_equal(res[r]["sum_i64"], v * 3)
        assert np.array_equal(res[r]["max_i64"], np.maximum(v, 2 * v))
        assert np.allclose(res[r]["sum_f64"], np.linspace(0.0, 1.0, 101) * 3, rtol=0, atol=1e-15)
        assert np.array_equal(res[r]["gathered"], np.stack([np.arange(6) + 10 * k for k in range(world)]))
        # every rank holds the whole BAM's result
        for i in range(len(ref["counters"])):
            if i not in SKIP:
                assert int(res[r]["counters"][i]) == int(ref["counters"][i]), (r, i)
        assert np.array_equal(res[r]["site_counts"], np.asarray(ref["site_counts"]))
        assert np.array_equal(res[r]["depth"], d_ref)
        assert int(res[r]["summaries"][:, 0].sum()) == 400_000
