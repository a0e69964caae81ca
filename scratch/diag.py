"""Step-by-step window-steps by zone (needs the -DAWV_DIAG build): python scratch/diag.py <lib> [pairs] [flags]"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AWV_HIP_LIB"] = sys.argv[1]
from allwave_amd import ffi, synth
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
data, offs, ids = synth.generate(256, 10000, 0.05, 2)
pairs = synth.all_pairs(256)[:n]
e = ffi.Engine(flags=flags | ffi.AWV_F_KEEP_ON_DEVICE)
e.set_sequences((data, offs))
res, _ = e.align_pairs((0, 5, 8, 2, 24, 1), pairs, want_cigars=False)
st = e.stats()
z = list(st.prof)[9:14]
names = ["no-pass sub-problems", "before passes can start", "margin zone (phase 1)", "phase 2", "trimmed rows"]
tot = sum(z) or 1
print(json.dumps({"pairs": n, "kernel_ms": round(st.kernel_ms, 1), "win_single": st.windows[0], "win_multi_sweeps": st.windows[1], "win_base": st.windows[2],
                  "win_base_multi_sweeps": st.windows[3], "cells": st.cell_steps, "multi_frac": round(st.multi_cell_steps / max(st.cell_steps, 1), 4),
                  "zones": {names[i]: [z[i], round(z[i] / tot, 4)] for i in range(5)}, "breakpoints": st.n_breakpoints, "base": st.n_base, "restarts": st.restarts}))
e.close()
