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
