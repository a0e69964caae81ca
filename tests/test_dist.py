"""The N>1 path on CPU: world_size-2 gloo processes shard the pair list, align their shards and
reduce counters; totals must equal the single-process run (no data-path collective)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_world(world, tmp_path):
    out = str(tmp_path / ("w%d.json" % world))
    worker = os.path.join(ROOT, "tests", "_dist_worker.py")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    if world == 1:
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        subprocess.check_call([sys.executable, worker, out], env=env, timeout=300)
    else:
        subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                               "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port",
                               str(29500 + world), worker, out], env=env, timeout=600)
    return json.load(open(out))


def test_shard_pairs_partition():
    from allwave_amd import dist as D, synth
    pairs = synth.all_pairs(9)
    for world in (1, 2, 3, 8):
        shards = [D.shard_pairs(pairs, r, world) for r in range(world)]
        allp = np.concatenate(shards)
        assert len(allp) == len(pairs)
        assert {tuple(p) for p in allp} == {tuple(p) for p in pairs}
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1


def test_two_ranks_equal_one(tmp_path):
    one = run_world(1, tmp_path)
    two = run_world(2, tmp_path)
    assert two["world"] == 2 and one["world"] == 1
    assert (two["bp"], two["n"], two["pen"]) == (one["bp"], one["n"], one["pen"])
    assert two["tmax"] == 1.5  # max over ranks of the per-rank timer
    # the gathered per-rank texts together are the single-process text
    assert two["lines"] == one["lines"] and len(one["lines"]) == one["n"]


def test_bench_launches_its_own_ranks():
    """A bare `python bench.py --gpus 2` (no WORLD_SIZE in the environment: the shape of the driver's
    command) starts two ranks itself and relays rank 0's line; --launch-check runs the launcher, the
    gloo rendezvous, the shard arithmetic and the reductions of the real run without a GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env, timeout=600)
    line = json.loads([ln for ln in out.decode().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2
    assert line["pairs_total"] == line["pairs_expected"] == 56 and line["pair_key_sum"] == line["pair_key_expected"]
    assert line["per_rank_pairs"] == [28.0, 28.0] and line["elapsed_max"] == 0.002
    # under a launcher, --gpus must agree with WORLD_SIZE
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                         env=dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"), capture_output=True, timeout=120)
    assert bad.returncode != 0 and b"WORLD_SIZE=3" in bad.stderr


def test_cost_balanced_shards_config5():
    """Config 5's pair list (512 prefixes of 1-50 kbp, -p tree:3:1:0.1): per-pair cost spans orders of
    magnitude.  The LPT partition keeps max/mean predicted shard cost <= 1.05 for N = 8, every pair
    lands on exactly one rank, and equal-cost lists come out strided."""
    from allwave_amd import dist as D, host as H, synth
    cfg = synth.CONFIGS["c5"]
    data, offs, ids = synth.generate(cfg["nseq"], cfg["length"], cfg["d"], cfg["seed"], mixed_lengths=cfg["mixed_lengths"])
    lens = (offs[1:] - offs[:-1]).astype(np.int64)
    seqs = [bytes(data[offs[i]:offs[i + 1]]) for i in range(cfg["nseq"])]
    plist = np.asarray(H.plan_pairs(ids, seqs, cfg["sparsify"]), dtype=np.int32)[:, :2]
    shard, cost = H.shard_assignment(plist, lens, "0,5,8,2,24,1", 8)
    assert cost.max() / cost.min() > 100  # the skew the partition has to absorb
    loads = np.array([cost[shard == r].sum() for r in range(8)])
    assert loads.max() / loads.mean() <= 1.05
    parts = [D.shard_pairs(plist, r, 8, lens=lens, scores=cfg["scores"]) for r in range(8)]
    assert sum(len(p) for p in parts) == len(plist)
    assert {tuple(p) for part in parts for p in part} == {tuple(p) for p in plist}
    for r in range(8):  # list order kept inside a shard
        idx = np.flatnonzero(shard == r)
        assert (parts[r] == plist[idx]).all()
    # equal costs (config 2 / 3): LPT is the strided shard
    eq = synth.all_pairs(12)
    sh, _ = H.shard_assignment(eq, np.full(12, 10000), "0,5,8,2,24,1", 8)
    assert (sh == np.arange(len(eq)) % 8).all()
