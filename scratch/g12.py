import sys, random, time, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from allwave_amd import ffi
from util import *
rng = random.Random(7)
seqs = []; pairs = []
for it in range(3000):
    s, t = random_pair(rng, 6000)
    seqs += [s, t]; pairs.append((len(seqs)-2, len(seqs)-1))
e = ffi.Engine()
e.set_sequences(seqs)
for rep in range(3):
    t0 = time.time(); res, cigs = e.align_pairs(DEFAULT_2P, pairs); t1 = time.time()
    st = e.stats()
    print("wall %.3f kernel %.3f launches %d h2d %.3f d2h %.3f cells %.3e" % (t1 - t0, st.kernel_ms / 1e3, st.launches, st.h2d_ms / 1e3, st.d2h_ms / 1e3, st.cell_steps), flush=True)
