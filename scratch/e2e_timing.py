"""Stage times of the end-to-end PAF path on config 2 (engine built with -DAWV_DEBUG_KNOBS, preloaded): 
LD_PRELOAD=scratch/ablibs/lib_dbg.so AWV_TIMING=1 python scratch/e2e_timing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from allwave_amd import synth, host as H
cfg = synth.CONFIGS["c2"]
data, offs, _ = synth.generate(cfg["nseq"], cfg["length"], cfg["d"], cfg["seed"])
seqs = [bytes(data[offs[i]:offs[i + 1]]) for i in range(cfg["nseq"])]
ids = ["s%05d" % i for i in range(cfg["nseq"])]
sc = ",".join(map(str, cfg["scores"]))
for rep in range(3):
    t0 = time.time()
    r = H.all_pairs_paf_count(ids, seqs, sc, orientation="forward", device=0, format_threads=16)
    print("== run %d: wall %.3f s, inside %.3f s, kernel %.1f ms, d2h %.1f ms, h2d %.1f ms" % (rep, time.time() - t0, r[2], r[3].kernel_ms, r[3].d2h_ms, r[3].h2d_ms), file=sys.stderr, flush=True)
