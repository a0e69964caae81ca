"""Measured balance of the cost-balanced (LPT) shards: config 5's pair list split for N GPUs, every shard run on this one
GPU in turn; kernel time per shard and max / mean (what an N-GPU run's slowest rank would lose).
usage: python scratch/shard_balance.py [N] [config]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from allwave_amd import ffi, synth, host as H
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
name = sys.argv[2] if len(sys.argv) > 2 else "c5"
cfg = synth.CONFIGS[name]
kw = {"mixed_lengths": cfg["mixed_lengths"]} if "mixed_lengths" in cfg else {}
data, offs, ids = synth.generate(cfg["nseq"], cfg["length"], cfg["d"], cfg["seed"], **kw)
lens = (offs[1:] - offs[:-1]).astype(np.int64)
pairs = np.asarray(H.plan_pairs(ids, [bytes(data[offs[i]:offs[i + 1]]) for i in range(cfg["nseq"])], cfg["sparsify"]), dtype=np.int32).reshape(-1, 2)
shard, cost = H.shard_assignment(pairs, lens, ",".join(map(str, cfg["scores"])), world)
e = ffi.Engine(flags=ffi.AWV_F_KEEP_ON_DEVICE)
e.set_sequences((data, offs))
ms, pc, npairs = [], [], []
for r in range(world):
    sub = np.ascontiguousarray(pairs[shard == r])
    res, _ = e.align_pairs(cfg["scores"], sub, want_cigars=False)
    st = e.stats()
    assert (res["status"] == 0).all()
    ms.append(round(st.kernel_ms, 1)); pc.append(float(cost[shard == r].sum())); npairs.append(len(sub))
e.close()
print(json.dumps({"config": name, "world": world, "pairs_per_shard": npairs, "kernel_ms_per_shard": ms, "kernel_ms_max_over_mean": round(max(ms) / (sum(ms) / len(ms)), 4),
                  "predicted_cost_max_over_mean": round(max(pc) / (sum(pc) / len(pc)), 4), "sum_ms": round(sum(ms), 1)}))
