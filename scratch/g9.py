import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from allwave_amd import ffi, synth
cfg = synth.CONFIGS["c2"]
data, offs, ids = synth.generate(cfg["nseq"], cfg["length"], cfg["d"], cfg["seed"])
nsub = 96
pairs = synth.all_pairs(nsub)
for rep in range(2):
    t0 = time.time(); e = ffi.Engine(); t1 = time.time()
    e.set_sequences((data[:offs[nsub]], offs[:nsub + 1])); t2 = time.time()
    res, cigs = e.align_pairs((0, 5, 8, 2, 24, 1), pairs); t3 = time.time()
    st = e.stats()
    res, cigs = e.align_pairs((0, 5, 8, 2, 24, 1), pairs); t4 = time.time()
    st2 = e.stats()
    e.close(); t5 = time.time()
    print("create %.3f setseq %.3f align1 %.3f (kernel %.3f d2h %.3f) align2 %.3f (kernel %.3f) close %.3f" % (t1 - t0, t2 - t1, t3 - t2, st.kernel_ms / 1e3, st.d2h_ms / 1e3, t4 - t3, st2.kernel_ms / 1e3, t5 - t4))
