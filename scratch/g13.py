import sys, random, time, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from allwave_amd import ffi
from util import *
rng = random.Random(3)
seqs = []; pairs = []
for it in range(2500):
    s = rand_seq(rng, rng.randint(2000, 9000)); t = mutate(s, rng.choice([0.02, 0.05, 0.1]), rng)
    seqs += [s, t]; pairs.append((len(seqs)-2, len(seqs)-1))
for scores in [(0, 6, 10, 3, 70, 2), (0, 5, 8, 2, 24, 1)]:
    for name, fl in (("one_wave", ffi.AWV_F_ONE_WAVE), ("four_waves", ffi.AWV_F_FOUR_WAVES)):
        e = ffi.Engine(flags=fl | ffi.AWV_F_KEEP_ON_DEVICE)
        e.set_sequences(seqs)
        res, _ = e.align_pairs(scores, pairs, want_cigars=False)
        st = e.stats()
        print(scores, name, "kernel_ms %.1f cells %.3e" % (st.kernel_ms, st.cell_steps), flush=True)
        e.close()
