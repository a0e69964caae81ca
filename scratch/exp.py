"""Experiment driver (GPU box): config-2-shaped runs with engine flags / pair subsets, prints kernel ms and stats.
usage: python scratch/exp.py [--flags N] [--pairs P] [--reps R] [--lib path] [--config c2] [--workgroups W]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--pairs", type=int, default=0)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--lib", default="")
ap.add_argument("--config", default="c2")
ap.add_argument("--workgroups", type=int, default=0)
ap.add_argument("--tag", default="")
a = ap.parse_args()
if a.lib:
    os.environ["AWV_HIP_LIB"] = a.lib
from allwave_amd import ffi, synth
cfg = synth.CONFIGS[a.config]
data, offs, _ = synth.generate(cfg["nseq"], cfg["length"], cfg["d"], cfg["seed"])
pairs = synth.all_pairs(cfg["nseq"])
if a.pairs:
    pairs = pairs[:a.pairs]
e = ffi.Engine(flags=a.flags | ffi.AWV_F_KEEP_ON_DEVICE, workgroups=a.workgroups)
e.set_sequences((data, offs))
for r in range(a.reps):
    res, _ = e.align_pairs(cfg["scores"], pairs, want_cigars=False)
    st = e.stats()
    bad = int((res["status"] != 0).sum())
    print(json.dumps({"tag": a.tag, "flags": a.flags, "rep": r, "pairs": len(pairs), "kernel_ms": round(st.kernel_ms, 2), "Mbp_s": round(st.aligned_bp / st.kernel_ms / 1e3, 1),
                      "cells": st.cell_steps, "multi_frac": round(st.multi_cell_steps / max(st.cell_steps, 1), 4), "deep_frac": round(st.deep_cell_steps / max(st.cell_steps, 1), 4), "clock_ghz": round(st.clock_cycles / max(st.clock_ticks, 1) * st.clock_tick_khz / 1e6, 3), "restarts": st.restarts, "win_single": st.windows[0], "win_multi": st.windows[1], "win_base": st.windows[2], "win_base_multi": st.windows[3],
                      "breakpoints": st.n_breakpoints, "bad": bad, "pen_sum": int(res["penalty"].sum()), "launches": st.launches}), flush=True)
e.close()
