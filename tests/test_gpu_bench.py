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


def _bench(*args, timeout=600):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_bench_line_on_a_small_shard():
    d = _bench("--reads", "3000000", "--steps", "2", "--warmup", "1", "--cpu-sample-reads", "1000000")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["vs_baseline"] is None and "workload" in d["config"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"]) and d["roofline"]["bound"] == "hbm"
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["counters_match_gpu"] is True
    assert d["config"]["members_inflated_per_step"] == d["config"]["bgzf_members"]      # one K1 visit per member for the whole MappingQC step
    assert d["roofline_scan"]["frac"] > 0 and "t_scan_ms" in d["roofline_scan"]


@pytest.mark.timeout(900)
def test_bench_spawns_two_ranks_and_reduces_their_counters():
    d = _bench("--gpus", "2", "--all-ranks-on-device0", "--backend", "gloo", "--reads", "2000000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["reads_per_gpu_per_step"] == 2000000 and "private copy per rank" in d["data"]
