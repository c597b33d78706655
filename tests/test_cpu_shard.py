"""One BAM sharded over several handles / ranks — host logic on CPU, no GPU:
ngsqc_plan_shard_fix (pure host function of libngsqc_hip.so) + the exchange protocol of ngs-bits_amd/dist.py against a
sequential model of the two order-dependent carries of the reference loop (running maximum read length behind
bases_trimmed, Statistics.cpp:428-429,565-568; "a paired read has been seen" behind bases_usable_no_overlap, :879,:1115),
in-process and as a world-size-2 gloo job."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ngsqc = __import__("importlib").import_module("ngs-bits_amd")

MODEL = r'''
import numpy as np

def sequential(recs):
    """recs: int array [n, 4] = (length, counted, passing, paired). Reference order semantics."""
    runmax, paired_seen, trimmed, no_overlap, n_counted = 0, False, 0, 0, 0
    for ln, counted, passing, paired in recs:
        if not counted:
            continue
        n_counted += 1
        runmax = max(runmax, ln)
        trimmed += runmax - ln
        if paired:
            paired_seen = True
        if passing and paired_seen:
            no_overlap += ln
    return dict(n=n_counted, trimmed=trimmed, no_overlap=no_overlap, gmax=runmax, paired=int(paired_seen))

class ShardModel:
    """What a shard handle does, on a slice of the record list (mirrors scan_mapping_partial / scan_mapping_finish)."""
    def __init__(self, recs, first_abs, exit_abs, n_slots, seed):
        self.recs, self.first_abs, self.exit_abs = recs, first_abs, exit_abs
        self.diff = np.random.default_rng(seed).integers(-3, 4, n_slots).astype(np.int32)
        self.finalized = False
    def scan_mapping_partial(self, mode, **kw):
        c = self.recs[self.recs[:, 1] != 0] if len(self.recs) else self.recs
        mx = int(c[:, 0].max()) if len(c) else 0
        fm = fp = -1
        for i, (ln, counted, passing, paired) in enumerate(self.recs):
            if counted and mx > 0 and ln == mx and fm < 0: fm = i
            if counted and paired and fp < 0: fp = i
        n = len(self.recs)
        return np.array([n, self.first_abs if n else -1, self.exit_abs if n else -1, mx, fm, fp], dtype=np.int64)
    def scan_mapping_finish(self, fix):
        run = int(fix.floor_max); fix_trim = fix_len = 0; n = s_len = usable = 0
        for i, (ln, counted, passing, paired) in enumerate(self.recs):
            if not counted: continue
            n += 1; s_len += ln
            if passing: usable += ln
            run = max(run, ln)
            if i < fix.trim_upto: fix_trim += fix.gmax - run
            if passing and i < fix.paired_upto: fix_len += ln
        c = np.zeros(1032, dtype=np.int64)
        c[0] = n; c[7] = n * fix.gmax - s_len - fix_trim; c[12] = (usable - fix_len) if fix.paired_end else 0
        c[24] = fix.gmax; c[25] = fix.paired_end; c[26] = 777
        return c, np.full(101, 0.5)
    def depth_device(self): return 0, self.diff.size
    def depth_diff(self): return self.diff
    def depth_diff_set(self, a): self.diff = np.asarray(a, dtype=np.int32).copy()
    def depth_finalize(self): self.finalized = True

def make(seed, n, n_shards):
    rng = np.random.default_rng(seed)
    lens = rng.integers(30, 100 + 10 * (seed % 7), n)
    late = rng.integers(0, n + 1)                       # reads of full length may only appear late in the file
    lens[late:] = np.maximum(lens[late:], rng.integers(30, 152, n - late))
    counted = (rng.random(n) < 0.95).astype(np.int64)
    passing = ((rng.random(n) < 0.8) & (counted == 1)).astype(np.int64)
    p0 = rng.integers(0, n + 2)                         # first paired read anywhere (or nowhere)
    paired = ((np.arange(n) >= p0) & (rng.random(n) < 0.7)).astype(np.int64)
    recs = np.stack([lens, counted, passing, paired], axis=1).astype(np.int64)
    cuts = np.sort(rng.integers(0, n + 1, n_shards - 1)) if n_shards > 1 else np.array([], dtype=np.int64)
    bounds = [0] + [int(x) for x in cuts] + [n]
    return recs, bounds
'''
exec(MODEL)


@pytest.mark.parametrize("n_shards", [1, 2, 3, 8])
def test_shard_protocol_matches_sequential_model(n_shards):
    for seed in range(60):
        n = [0, 1, 5, 40, 300][seed % 5]
        recs, bounds = make(seed, n, n_shards)
        shards = [ShardModel(recs[bounds[i]:bounds[i + 1]], 1000 + 10 * bounds[i], 1000 + 10 * bounds[i + 1], 50, 7 + i) for i in range(n_shards)]
        want = sequential(recs)
        want_depth = np.sum(np.stack([s.diff for s in shards]).astype(np.int64), axis=0).astype(np.int32)
        counters, gc, summaries = ngsqc.scan_mapping_sharded_local(shards, ngsqc.MODE_WGS)
        assert counters[0] == want["n"] and counters[7] == want["trimmed"], (seed, n_shards, counters[7], want)
        assert counters[12] == want["no_overlap"] and counters[24] == want["gmax"] and counters[25] == want["paired"], (seed, n_shards)
        assert counters[26] == 777                          # roi_bases is not additive
        assert np.array_equal(shards[0].diff, want_depth) and shards[0].finalized
        assert summaries.shape == (n_shards, 6)


def test_chain_mismatch_between_shards_is_an_error():
    s = np.array([[10, 100, 500, 100, 3, -1], [0, -1, -1, 0, -1, -1], [7, 500, 900, 150, 0, 0]], dtype=np.int64)
    fix = ngsqc.plan_shard_fix(s, 2)                         # the empty middle shard is skipped by the chain check
    assert (fix.gmax, fix.floor_max, fix.trim_upto, fix.paired_upto, fix.paired_end) == (150, 100, 0, 0, 1)
    fix0 = ngsqc.plan_shard_fix(s, 0)
    assert (fix0.trim_upto, fix0.paired_upto) == (10, 10)    # everything in shard 0 precedes the first 150-base / paired read
    s[2, 1] = 501
    with pytest.raises(ngsqc.NgsqcError) as e:
        ngsqc.plan_shard_fix(s, 0)
    assert "record chain" in str(e.value)


_WORKER = MODEL + r'''
import importlib, os, sys
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
ngsqc = importlib.import_module("ngs-bits_amd")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
for seed in range(12):
    recs, bounds = make(seed, [3, 50, 400][seed % 3], world)
    shards = [ShardModel(recs[bounds[i]:bounds[i + 1]], 1000 + 10 * bounds[i], 1000 + 10 * bounds[i + 1], 64, 7 + i) for i in range(world)]
    want = sequential(recs)
    want_depth = np.sum(np.stack([s.diff for s in shards]).astype(np.int64), axis=0).astype(np.int32)
    counters, gc, summaries = ngsqc.scan_mapping_sharded(shards[rank], ngsqc.MODE_WGS)
    assert counters[0] == want["n"] and counters[7] == want["trimmed"] and counters[12] == want["no_overlap"], (rank, seed)
    assert counters[24] == want["gmax"] and counters[25] == want["paired"] and counters[26] == 777, (rank, seed)
    assert np.array_equal(shards[rank].diff, want_depth) and shards[rank].finalized, (rank, seed)
    assert abs(gc[3] - 0.5 * world) < 1e-12
dist.barrier()
dist.destroy_process_group()
print("ok", rank)
'''


def test_sharded_scan_protocol_gloo_world2(tmp_path):
    """The N>1 path of a single sharded BAM: all-gather of the summaries, plan, all-reduce of counters / gc / difference array."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err[-3000:]
        assert "ok" in out
