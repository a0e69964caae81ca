import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from allwave_amd import ffi, synth
data, offs, ids = synth.generate(256, 10000, 0.05, 2)
pairs = synth.all_pairs(256)
e = ffi.Engine(workgroups=4096, flags=ffi.AWV_F_KEEP_ON_DEVICE | ffi.AWV_F_ONE_WAVE)
e.set_sequences((data, offs))
n = 32768
ts = []
for rep in range(4):
    res,_ = e.align_pairs((0,5,8,2,24,1), pairs[:n], want_cigars=False)
    ts.append(e.stats().kernel_ms)
st = e.stats()
print("kernel_ms", " ".join("%.0f" % t for t in ts), "| cells %.4e bp %d bad %d ovscans %d ext %.3e" % (st.cell_steps, st.n_breakpoints, int((res["status"]!=0).sum()), st.overlap_scans, st.extend_steps))
