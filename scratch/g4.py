import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from allwave_amd import ffi, synth
data, offs, ids = synth.generate(256, 10000, 0.05, 2)
pairs = synth.all_pairs(256)
scores = (0,5,8,2,24,1)
e = ffi.Engine(workgroups=int(sys.argv[1]) if len(sys.argv)>1 else 1024, flags=ffi.AWV_F_KEEP_ON_DEVICE)
e.set_sequences((data, offs))
n = int(sys.argv[2]) if len(sys.argv)>2 else 4096
res,_ = e.align_pairs(scores, pairs[:n], want_cigars=False)
st = e.stats()
print("kernel_ms %.1f pairs/s %.1f cells %d" % (st.kernel_ms, n/(st.kernel_ms*1e-3), st.cell_steps))
