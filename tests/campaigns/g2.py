import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from allwave_amd import ffi, synth
from oracle import oracle as O
data, offs, ids = synth.generate(256, 10000, 0.05, 2)
pairs = synth.all_pairs(256)
scores = (0,5,8,2,24,1)
for wg in (256, 512, 1024):
    e = ffi.Engine(workgroups=wg, flags=ffi.AWV_F_KEEP_ON_DEVICE)
    e.set_sequences((data, offs))
    for n in (2048, 8192):
        sub = pairs[:n]
        t0=time.time(); res,_ = e.align_pairs(scores, sub, want_cigars=False); t1=time.time()
        st = e.stats()
        ok = int((res['status']==0).sum())
        print("wg",wg,"n",n,"ok",ok,"wall %.3f kernel_ms %.1f cells %.3e cells/s %.3e pairs/s %.1f Mbp/s %.1f scratchGB %.1f bp %d base %d ov %d ext %.3e" % (
            t1-t0, st.kernel_ms, st.cell_steps, st.cell_steps/(st.kernel_ms*1e-3), n/(st.kernel_ms*1e-3), st.aligned_bp/(st.kernel_ms*1e-3)/1e6, st.scratch_bytes/2**30, st.n_breakpoints, st.n_base, st.overlap_scans, st.extend_steps), flush=True)
    e.close()
# parity on a sample vs oracle
e = ffi.Engine()
e.set_sequences((data, offs))
sub = pairs[::997][:64]
res, cigs = e.align_pairs(scores, sub)
secs, ores, ost, _ = O.all_pairs(data, offs, sub, scores, nthreads=8)
import zlib
bad=0
al = O.Aligner(scores)
for i,(a,b) in enumerate(sub):
    pen,cg = al.align(bytes(data[offs[a]:offs[a+1]]), bytes(data[offs[b]:offs[b+1]]))
    if pen!=res['penalty'][i] or cg!=cigs[i]: bad+=1
print("sample parity bad", bad, "of", len(sub), "cpu secs(8thr) %.2f" % secs, "cpu cells", ost.cell_steps)
