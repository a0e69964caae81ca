import sys, os, time, random
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from allwave_amd import ffi
from util import *
rng = random.Random(5)
a = rand_seq(rng, 30000)
b = mutate(a, 0.05, rng)[:2000]
c = mutate(a, 0.10, rng)
e = ffi.Engine(flags=ffi.AWV_F_KEEP_ON_DEVICE)
e.set_sequences([a, b, c])
for pr in ([(1, 0)], [(0, 2)], [(1, 0), (0, 1), (0, 2), (2, 0)]):
    res, _ = e.align_pairs((0, 5, 8, 2, 24, 1), pr, want_cigars=False)
    st = e.stats()
    print(pr, "status", list(res["status"]), "penalty", list(res["penalty"]), "kernel_ms %.1f cells %.3e launches %d" % (st.kernel_ms, st.cell_steps, st.launches), flush=True)
