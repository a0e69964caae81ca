"""Cell-steps per second of 2 %-divergent equal-length pairs just below and just above the length at which the top BiWFA level no
longer fits the LDS staging of the packed sequences (four waves per pair, 32-bit rows): python scratch/lenscan.py [L ...]"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from allwave_amd import ffi, synth
for L in [int(x) for x in sys.argv[1:]] or [60000, 70000]:
    data, offs, _ = synth.generate(48, L, 0.02, 11)
    pairs = synth.all_pairs(48)[:2048]
    e = ffi.Engine(flags=ffi.AWV_F_KEEP_ON_DEVICE)
    e.set_sequences((data, offs))
    for rep in range(2):
        res, _ = e.align_pairs((0, 5, 8, 2, 24, 1), pairs, want_cigars=False)
        st = e.stats()
    print(json.dumps({"L": L, "pairs": len(pairs), "kernel_ms": round(st.kernel_ms, 1), "Gcells_per_s": round(st.cell_steps / st.kernel_ms / 1e6, 2),
                      "multi_frac": round(st.multi_cell_steps / st.cell_steps, 3), "bad": int((res["status"] != 0).sum())}), flush=True)
    e.close()
