"""Kernel-counted cell-steps against the oracle's on a config-2 sample: python scratch/cells_check.py [npairs] [stride]"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from allwave_amd import ffi, synth
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 677
data, offs, _ = synth.generate(256, 10000, 0.05, 2)
sample = synth.all_pairs(256)[::stride][:n]
out = {}
for fast in (True, False):
    secs, ores, ost, _ = O.all_pairs(data, offs, sample, (0, 5, 8, 2, 24, 1), nthreads=16, fast_overlap=fast)
    out["oracle_fast" if fast else "oracle_plain"] = int(ost.cell_steps)
    out["oracle_secs_%d" % fast] = round(secs, 1)
for name, flags in (("passes", ffi.AWV_F_ONE_WAVE), ("nodeep", ffi.AWV_F_ONE_WAVE | ffi.AWV_F_NO_DEEP), ("single", ffi.AWV_F_ONE_WAVE | ffi.AWV_F_SINGLE_STEP)):
    e = ffi.Engine(flags=flags)
    e.set_sequences((data, offs))
    res, _ = e.align_pairs((0, 5, 8, 2, 24, 1), sample, want_cigars=False)
    st = e.stats()
    out[name] = {"cells": int(st.cell_steps), "restarts": int(st.restarts), "rel_fast": round(st.cell_steps / out["oracle_fast"] - 1, 5), "rel_plain": round(st.cell_steps / out["oracle_plain"] - 1, 5)}
    e.close()
print(json.dumps(out))
