"""Configs 4 / 5 at full size in ONE awv_align_pairs call: kernel time, oracle-free invariants on every pair, and
an evenly spaced sample against the oracle (bounded).  usage: python scratch/c45.py c5 [oracle_pairs] [flags]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from allwave_amd import ffi, synth, host as H
from oracle import oracle as O
name = sys.argv[1]
nor = int(sys.argv[2]) if len(sys.argv) > 2 else 64
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cfg = synth.CONFIGS[name]
kw = {"mixed_lengths": cfg["mixed_lengths"]} if "mixed_lengths" in cfg else {}
data, offs, ids = synth.generate(cfg["nseq"], cfg["length"], cfg["d"], cfg["seed"], **kw)
lens = (offs[1:] - offs[:-1]).astype(np.int64)
seqs = [bytes(data[offs[i]:offs[i + 1]]) for i in range(cfg["nseq"])]
pairs = np.asarray(H.plan_pairs(ids, seqs, cfg["sparsify"]), dtype=np.int32).reshape(-1, 2)
del seqs
sel = sys.argv[4] if len(sys.argv) > 4 else "all"   # all | rows16 | rows32 | shorttext (tlen < 32760 <= plen) | shortpattern
ql, tl = lens[pairs[:, 0]], lens[pairs[:, 1]]
if sel == "rows16":
    pairs = pairs[(ql < 32760) & (tl < 32760)]
elif sel == "rows32":
    pairs = pairs[(ql >= 32760) | (tl >= 32760)]
elif sel == "shorttext":
    pairs = pairs[(ql >= 32760) & (tl < 32760)]
elif sel == "shortpattern":
    pairs = pairs[(ql < 32760) & (tl >= 32760)]
stride = int(os.environ.get("C45_STRIDE", "1"))  # every stride-th pair of the selection (profiling runs)
pairs = np.ascontiguousarray(pairs[::stride])
e = ffi.Engine(flags=flags | ffi.AWV_F_KEEP_ON_DEVICE)
e.set_sequences((data, offs))
t0 = time.time()
res, _ = e.align_pairs(cfg["scores"], pairs, want_cigars=False)
wall = time.time() - t0
st = e.stats()
ok = (res["status"] == 0) & (res["q_end"] == lens[pairs[:, 0]]) & (res["t_end"] == lens[pairs[:, 1]]) & \
     (res["num_matches"] + res["num_mismatches"] + res["num_ins"] + res["num_del"] == res["cigar_len"])
bp = int(lens[pairs[:, 0]].sum())
esz = 2  # bytes per row element where both lengths fit 16 bits; quoted per cell-step below at the 16-bit size
print(json.dumps({"config": name, "select": sel, "pairs": len(pairs), "failed_invariants": int((~ok).sum()), "wall_s": round(wall, 2), "kernel_ms": round(st.kernel_ms, 1),
                  "launches": st.launches, "Mbp_s_kernel": round(bp / st.kernel_ms / 1e3, 2), "cell_steps": st.cell_steps,
                  "cell_steps_per_s": st.cell_steps / (st.kernel_ms * 1e-3), "multi_frac": round(st.multi_cell_steps / max(st.cell_steps, 1), 4),
                  "restarts": st.restarts, "status": {int(k): int(v) for k, v in zip(*np.unique(res["status"], return_counts=True))}}), flush=True)
e.close()
# oracle sample
idx = np.unique(np.linspace(0, len(pairs) - 1, nor).astype(np.int64))
sub = np.ascontiguousarray(pairs[idx])
e = ffi.Engine(flags=flags)
e.set_sequences((data, offs))
gres, gc = e.align_pairs(cfg["scores"], sub)
e.close()
secs, ores, _, _ = O.all_pairs(data, offs, sub, cfg["scores"], nthreads=min(16, len(os.sched_getaffinity(0))), fast_overlap=True)
bad = 0
for i in range(len(sub)):
    if gres["status"][i] != 0 or gres["penalty"][i] != ores["penalty"][i] or O.fnv1a(gc[i]) != int(ores["cigar_hash"][i]) or gres["penalty"][i] != res["penalty"][idx[i]]:
        bad += 1
print(json.dumps({"config": name, "oracle_sample": len(sub), "mismatches": bad, "oracle_s": round(secs, 1)}), flush=True)
