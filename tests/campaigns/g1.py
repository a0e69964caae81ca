import sys, random, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from allwave_amd import ffi
from oracle import oracle as O
from util import *
e = ffi.Engine()
rng = random.Random(11)
# KATs
ref = b"ATCG"*25
q = bytearray(ref); q[10]=ord('G'); q[20]=ord('C'); del q[30]; q.insert(40, ord('A'))
r, cg = e.align_one(DEFAULT_2P, bytes(q), ref); print("KAT1", r['status'], r['penalty'], rle(cg))
bad = 0; total = 0
for scores in PENALTY_SETS:
    seqs = []; pairs = []
    for it in range(200):
        s, t = random_pair(rng, 1500)
        seqs += [s, t]; pairs.append((len(seqs)-2, len(seqs)-1))
    e.set_sequences(seqs)
    t0 = time.time(); res, cigs = e.align_pairs(scores, pairs); t1 = time.time()
    al = O.Aligner(scores)
    nb = 0
    for i, (a, b) in enumerate(pairs):
        pen, cg = al.align(seqs[a], seqs[b])
        total += 1
        if res['status'][i] != 0 or res['penalty'][i] != pen or cigs[i] != cg:
            nb += 1
            if nb <= 3: print("MISMATCH", scores, i, len(seqs[a]), len(seqs[b]), res['status'][i], res['penalty'][i], pen, rle(cigs[i] or b'')[:80], rle(cg)[:80])
    bad += nb
    st = e.stats()
    print(scores, "bad", nb, "time %.3f kernel_ms %.2f cells %d bp %d base %d ov %d" % (t1-t0, st.kernel_ms, st.cell_steps, st.n_breakpoints, st.n_base, st.overlap_scans))
print("TOTAL bad", bad, "of", total)
