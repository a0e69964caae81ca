"""Cycle-stamp profile (needs the -DAWV_PROF build): python scratch/prof.py <lib> [flags] [pairs]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AWV_HIP_LIB"] = sys.argv[1]
from allwave_amd import ffi, synth
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
data, offs, ids = synth.generate(256, 10000, 0.05, 2)
pairs = synth.all_pairs(256)[:n]
e = ffi.Engine(flags=flags | ffi.AWV_F_KEEP_ON_DEVICE)
e.set_sequences((data, offs))
res, _ = e.align_pairs((0, 5, 8, 2, 24, 1), pairs, want_cigars=False)
st = e.stats()
p = list(st.prof)
tot = p[0] or 1
names = ["total", "bi_compute", "bi_barrier", "bi_finalize", "overlap", "base_steps", "backtrace", "emit"]
print("flags %d pairs %d kernel_ms %.1f  multi_frac %.3f  windows single %d multi-passes %d base %d base-multi-passes %d" % (flags, n, st.kernel_ms, st.multi_cell_steps / max(st.cell_steps, 1), st.windows[0], st.windows[1], st.windows[2], st.windows[3]))
print("  phases (%% of wave-0 cycles): " + "  ".join("%s %.1f%%" % (names[i], 100.0 * p[i] / tot) for i in range(1, 8)))
crn = ["load+wait", "dp", "extend", "store", "reduce"]
wsteps = st.windows[0] + st.windows[2] + 5 * (st.windows[1] + st.windows[3])
print("  inside the step code: " + "  ".join("%s %.1f%% (%.0f cyc/window-step)" % (crn[i], 100.0 * p[9 + i] / tot, p[9 + i] / max(wsteps, 1)) for i in range(5)),
      " | total cycles per window-step %.0f" % (tot / max(wsteps, 1)))
e.close()
