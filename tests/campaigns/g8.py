"""Stress campaign: many random pairs of varied shape, GPU vs the oracle's thread-pool driver
(penalty, length, op counts and FNV-1a of the op bytes)."""
import sys, random, time, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
import numpy as np
from allwave_amd import ffi
from oracle import oracle as O
from util import *
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
maxlen = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
rng = random.Random(seed)
e = ffi.Engine(flags=int(os.environ.get("G8_FLAGS", "0")))
bad = total = 0
sets = PENALTY_SETS + [(0, 6, 10, 3, 70, 2), (0, 1, 1, 1)]
for scores in sets:
    seqs = []; pairs = []
    for it in range(npairs):
        kind = rng.random()
        if kind < 0.6:
            s, t = random_pair(rng, maxlen)
        elif kind < 0.8:   # very different lengths: prefix / suffix / infix of a mutated copy
            n = rng.randint(50, maxlen)
            s = rand_seq(rng, n)
            t = mutate(s, rng.choice([0.0, 0.02, 0.1]), rng)
            a = rng.randint(0, len(t) // 2); b = rng.randint(a, len(t))
            t = t[a:b] if rng.random() < 0.7 else t[:b]
            if rng.random() < 0.5: s, t = t, s
        elif kind < 0.9:   # low-complexity / repeats
            unit = rand_seq(rng, rng.randint(1, 6))
            s = (unit * (maxlen // len(unit)))[:rng.randint(10, maxlen)]
            t = mutate(s, rng.choice([0.01, 0.05]), rng)
        else:              # unrelated
            s = rand_seq(rng, rng.randint(1, 400)); t = rand_seq(rng, rng.randint(1, 400))
        seqs += [s, t]; pairs.append((len(seqs)-2, len(seqs)-1))
    e.set_sequences(seqs)
    t0 = time.time(); res, cigs = e.align_pairs(scores, pairs); t1 = time.time()
    data = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64); offs[1:] = np.cumsum([len(x) for x in seqs])
    secs, ores, st, _ = O.all_pairs(data, offs, np.array(pairs, dtype=np.int32), scores, nthreads=16)
    nb = 0
    for i in range(len(pairs)):
        total += 1
        ok = (res['status'][i] == 0 and ores['status'][i] == 0 and res['penalty'][i] == ores['penalty'][i]
              and res['cigar_len'][i] == ores['cigar_len'][i] and O.fnv1a(cigs[i]) == int(ores['cigar_hash'][i]))
        if not ok:
            nb += 1
            if nb <= 3: print("MISMATCH", scores, i, len(seqs[2*i]), len(seqs[2*i+1]), res['status'][i], res['penalty'][i], ores['penalty'][i])
    bad += nb
    print(scores, "bad", nb, "gpu %.2fs oracle %.2fs" % (t1 - t0, secs), flush=True)
print("TOTAL bad", bad, "of", total)
