"""Config-3-shaped soak: 4096 x 10 kbp sequence set, a 300k-pair slice of the all-pairs list (several launch
batches), invariants only: all complete, CIGAR lengths consistent, symmetric penalties on a sample."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from allwave_amd import ffi, synth
cfg = synth.CONFIGS["c3"]
t0 = time.time()
data, offs, ids = synth.generate(cfg["nseq"], cfg["length"], cfg["d"], cfg["seed"])
print("generated %d seqs in %.1f s" % (cfg["nseq"], time.time() - t0), flush=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
rng = np.random.default_rng(3)
qi = rng.integers(0, cfg["nseq"], n); ti = (qi + 1 + rng.integers(0, cfg["nseq"] - 1, n)) % cfg["nseq"]
pairs = np.stack([qi, ti], axis=1).astype(np.int32)
pairs[1::2] = pairs[0::2][:len(pairs[1::2]), ::-1]   # odd entries = the swapped even pair
e = ffi.Engine(flags=ffi.AWV_F_KEEP_ON_DEVICE, max_arena_bytes=2 << 30)
e.set_sequences((data, offs))
t0 = time.time()
res, _ = e.align_pairs(cfg["scores"], pairs, want_cigars=False)
st = e.stats()
print("pairs %d launches %d wall %.1f s kernel %.1f s  pairs/s %.0f  Mbp/s %.1f" % (n, st.launches, time.time() - t0, st.kernel_ms / 1e3, n / (st.kernel_ms / 1e3), st.aligned_bp / (st.kernel_ms / 1e3) / 1e6))
assert (res["status"] == 0).all(), int((res["status"] != 0).sum())
ql = (offs[pairs[:, 0] + 1] - offs[pairs[:, 0]]).astype(np.int64); tl = (offs[pairs[:, 1] + 1] - offs[pairs[:, 1]]).astype(np.int64)
assert (res["q_end"] == ql).all() and (res["t_end"] == tl).all()
assert (res["num_matches"] + res["num_mismatches"] + res["num_ins"] + res["num_del"] == res["cigar_len"]).all()
m = len(pairs[1::2])
assert (res["penalty"][0:2 * m:2] == res["penalty"][1:2 * m:2]).all(), "penalty(a,b) != penalty(b,a)"
print("invariants ok; mean penalty %.1f" % res["penalty"].mean())
