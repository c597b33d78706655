"""bench.py contract on a GPU box: the JSON line, and the N > 1 launch path. `python bench.py --gpus 2` must start two ranks itself; RCCL refuses two
ranks on ONE device ("Duplicate GPU detected", tried on the round-2 box), so on a single-GPU box the two ranks share the device and exchange their
counter vectors over gloo - the rank spawning, the shared image, the per-step all-reduce and the max-over-ranks timing are the same code as with nccl."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(*args, timeout=600, env=None):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_bench_line_on_a_small_shard():
    d = _bench("--reads", "3000000", "--steps", "2", "--warmup", "1", "--cpu-sample-reads", "1000000", env={"NGSQC_BENCH_ONT_READS": "6000", "NGSQC_BENCH_FLAVOR_READS": "1500000"})
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["vs_baseline"] is None and "workload" in d["config"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"]) and d["roofline"]["bound"] == "hbm"
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["counters_match_gpu"] is True
    assert d["config"]["members_inflated_per_step"] == d["config"]["bgzf_members"]      # one K1 visit per member for the whole MappingQC step
    assert d["roofline_scan"]["frac"] > 0 and "t_scan_ms" in d["roofline_scan"]
    # BASELINE.json configs[2] and configs[4] ride in the same line: BedCoverage / BedLowCoverage (with and without -min_baseq 20) over the same
    # resident image, and the ONT shard - each with its value, scan-stage roofline, CPU baseline and parity of the sample
    assert set(d["tools"]) == {"bedcoverage", "bedlowcoverage", "bedlowcoverage_baseq20"}
    for leg in list(d["tools"].values()) + [d["ont"]]:
        assert "error" not in leg, leg
        assert leg["value"] > 0 and leg["ms_per_step"] > 0 and leg["steps"] >= 3 and leg["roofline_scan"]["frac"] > 0
        assert leg["cpu_baseline"]["kind"] == "port" and leg["counters_match_gpu"] is True
    assert all(leg["reads_per_step"] == d["config"]["reads_per_gpu_per_step"] for leg in d["tools"].values())
    assert d["ont"]["reads_per_step"] == 6000 and "configs[4]" in d["ont"]["workload"]
    # the realistic token mix (reference-derived SEQ, 40-level QUAL) rides in the same line
    fl = d["flavors"]["flavor5_refseq_40level_qual"]
    assert "error" not in fl and fl["value"] > 0 and fl["reads_per_step"] == 1500000 and fl["counters_match_gpu"] is True and fl["inflate_stage_unpipelined_ms"] > 0


@pytest.mark.timeout(900)
def test_bench_spawns_two_ranks_and_reduces_their_counters():
    d = _bench("--gpus", "2", "--all-ranks-on-device0", "--backend", "gloo", "--reads", "2000000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["reads_per_gpu_per_step"] == 2000000
    # one BAM per GPU with its own seed (the host has room for two 0.2 GB images); the reduce is checked against a second channel on every rank
    assert "seed + rank" in d["data"] and d["collective"]["one_bam_per_gpu_with_its_own_seed"] is True and d["collective"]["distinct_inputs"] == 2
    assert d["collective"]["allreduce_matches_gathered_sum_per_rank"] == [True, True]


@pytest.mark.timeout(900)
def test_bench_two_ranks_on_one_shared_image():
    """the fallback when the host cannot hold one image per rank: rank 0 generates, the others map it from /dev/shm; identical inputs, the same reduce"""
    d = _bench("--gpus", "2", "--all-ranks-on-device0", "--backend", "gloo", "--reads", "2000000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", env={"NGSQC_BENCH_SHARED_IMAGE": "1"})
    assert d["n_gpus"] == 2 and "private copy per rank" in d["data"] and d["collective"]["distinct_inputs"] == 1
    assert d["collective"]["allreduce_matches_gathered_sum_per_rank"] == [True, True]


@pytest.mark.timeout(1200)
def test_bench_eight_ranks_rehearsal():
    """`bench.py --gpus 8` end to end without eight GPUs - eight ranks (gloo) share this box's one device: rank spawning, ONE generated image shared through
    /dev/shm and mapped by eight ranks, eight handles, the per-step all-reduce checked against a second channel on every rank, the one-BAM-over-eight-shards leg, the
    barriers and max-over-ranks timing, the budget of the run. (RCCL itself refuses two ranks on one device; its call sequence is what test_gpu_shard's world-of-one runs.)"""
    d = _bench("--gpus", "8", "--all-ranks-on-device0", "--backend", "gloo", "--reads", "1500000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", timeout=1100, env={"NGSQC_BENCH_SHARED_IMAGE": "1"})
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["reads_per_gpu_per_step"] == 1500000
    assert d["collective"]["allreduce_matches_gathered_sum_per_rank"] == [True] * 8 and d["collective"]["distinct_inputs"] == 1
    sb = d["single_bam"]
    assert "error" not in sb and sb["scaling"] == "strong" and sb["counters_match_one_gpu_job"] is True and sb["members_inflated_per_step"] >= sb["bgzf_members"]
    assert d["budget"]["total_s"] > 0 and len(d["budget"]["phases"]) >= 4
