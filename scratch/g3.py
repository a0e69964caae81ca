import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np
from allwave_amd import ffi, synth
data, offs, ids = synth.generate(256, 10000, 0.05, 2)
pairs = synth.all_pairs(256)
scores = (0,5,8,2,24,1)
names = ["total","bi_compute","bi_barrier","bi_finalize","overlap","base_steps","backtrace","emit","passes"]
for wg in [int(x) for x in (sys.argv[1:] or ["1024"])]:
    e = ffi.Engine(workgroups=wg, flags=ffi.AWV_F_KEEP_ON_DEVICE)
    e.set_sequences((data, offs))
    n = 4096
    res,_ = e.align_pairs(scores, pairs[:n], want_cigars=False)
    st = e.stats()
    tot = st.prof[0] or 1
    print("wg",wg,"kernel_ms %.1f pairs/s %.1f cells/s %.3e" % (st.kernel_ms, n/(st.kernel_ms*1e-3), st.cell_steps/(st.kernel_ms*1e-3)))
    print("  " + "  ".join("%s %.1f%%" % (names[i], 100.0*st.prof[i]/tot) for i in range(1,8)), " passes/pair %.0f cycles/pass %.0f" % (st.prof[8]/n, (st.prof[1]+st.prof[2]+st.prof[3])/max(st.prof[8],1)))
    crn=["load","alu","extend","store","reduce"]
    crt=sum(st.prof[9:14]) or 1
    print("  compute_row split (incl. base): " + "  ".join("%s %.1f%%" % (crn[i], 100.0*st.prof[9+i]/crt) for i in range(5)), " cr_total/total %.1f%%" % (100.0*crt/tot))
    e.close()
