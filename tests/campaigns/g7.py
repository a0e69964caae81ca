import sys, random, time, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from allwave_amd import ffi
from oracle import oracle as O
from util import *
e = ffi.Engine()
rng = random.Random(77)
bad = total = 0
for scores in [(0, 6, 10, 3, 70, 2), (0, 80, 5, 2), (0, 1, 1, 1), (0, 3, 90, 1), (0, 2, 4, 2, 100, 1), (0, 9, 3, 7, 30, 5)]:
    seqs = []; pairs = []
    for it in range(60):
        s, t = random_pair(rng, 4000)
        seqs += [s, t]; pairs.append((len(seqs)-2, len(seqs)-1))
    e.set_sequences(seqs)
    res, cigs = e.align_pairs(scores, pairs)
    al = O.Aligner(scores)
    nb = 0
    for i, (a, b) in enumerate(pairs):
        pen, cg = al.align(seqs[a], seqs[b])
        total += 1
        if res['status'][i] != 0 or res['penalty'][i] != pen or cigs[i] != cg:
            nb += 1
            if nb <= 3: print("MISMATCH", scores, i, len(seqs[a]), len(seqs[b]), res['status'][i], res['penalty'][i], pen)
    bad += nb
    st = e.stats()
    print(scores, "bad", nb, "kernel_ms %.2f bp %d base %d" % (st.kernel_ms, st.n_breakpoints, st.n_base), flush=True)
print("TOTAL bad", bad, "of", total)
