"""FORCE_INT32 runs of a library variant against the default library's 16-bit rows (config-2-shaped sample): status codes and mismatches."""
import os, sys, json, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from allwave_amd import ffi, synth
cfg = synth.CONFIGS["c2"]
data, offs, _ = synth.generate(cfg["nseq"], cfg["length"], cfg["d"], cfg["seed"])
pairs = synth.all_pairs(cfg["nseq"])[:int(sys.argv[1]) if len(sys.argv) > 1 else 2048]
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 0
out = {}
for name, flags in (("i32", ffi.AWV_F_FORCE_INT32), ("i32_single", ffi.AWV_F_FORCE_INT32 | ffi.AWV_F_SINGLE_STEP)):
    e = ffi.Engine(flags=flags | extra | ffi.AWV_F_KEEP_ON_DEVICE)
    e.set_sequences((data, offs))
    res, _ = e.align_pairs(cfg["scores"], pairs, want_cigars=False)
    st = e.stats()
    e.close()
    out[name] = res
    print(name, json.dumps({"kernel_ms": round(st.kernel_ms, 1), "status": dict(collections.Counter(int(x) for x in res["status"])), "multi_frac": round(st.multi_cell_steps / max(1, st.cell_steps), 3), "restarts": st.restarts}), flush=True)
a, b = out["i32"], out["i32_single"]
bad = [f for f in a.dtype.names if (a[f] != b[f]).any()]
print("fields differing:", bad, "pairs differing in penalty:", int((a["penalty"] != b["penalty"]).sum()))
