import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from allwave_amd import ffi, synth
data, offs, ids = synth.generate(256, 10000, 0.05, 2)
pairs = synth.all_pairs(256)
e = ffi.Engine(workgroups=4096, flags=ffi.AWV_F_KEEP_ON_DEVICE | ffi.AWV_F_ONE_WAVE)
e.set_sequences((data, offs))
n = 32768
res,_ = e.align_pairs((0,5,8,2,24,1), pairs[:n], want_cigars=False)
st = e.stats()
print("kernel_ms %.1f pairs/s %.1f cells %.4e status!=0: %d mean penalty %.1f bp %d" % (st.kernel_ms, n/(st.kernel_ms*1e-3), st.cell_steps, int((res["status"]!=0).sum()), res["penalty"].mean(), st.n_breakpoints))
