import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from allwave_amd import ffi, synth
data, offs, ids = synth.generate(256, 10000, 0.05, 2)
pairs = synth.all_pairs(256)
e = ffi.Engine(workgroups=4096, flags=ffi.AWV_F_KEEP_ON_DEVICE)
e.set_sequences((data, offs))
n = 4096
res,_ = e.align_pairs((0,5,8,2,24,1), pairs[:n], want_cigars=False)
st = e.stats()
p = list(st.prof)
print("pairs", n, "cells/pair %.3e" % (st.cell_steps/n), "bp/pair %.1f base/pair %.1f passes/pair %.1f" % (st.n_breakpoints/n, st.n_base/n, p[8]/n))
print("per pair: counters", ["%.1f" % (x/n) for x in p[9:14]])
