"""Full-size parity campaign: the pairs of a BASELINE.json config at its REAL size (default c2: 256 x
10 kbp, all 65,280 directed pairs, scores 0,5,8,2,24,1; c3 / c4 / c5: the config's own read set and
its own pair list -- c4 and c5 through the host planner's sparsifier -- as far as the oracle-seconds
budget reaches) through the C ABI on the GPU against the CPU oracle's thread-pool driver: penalty,
op-byte length, M/X/I/D counts and FNV-1a of the CIGAR op bytes, pair by pair.  (The oracle runs in
its baseline mode, which tests/test_oracle.py proves equal to its plain WFA2-order search.)
Hand-run: `gpurun -- python tests/campaigns/config_full.py [config] [first] [count] [oracle_seconds] [chunk]`;
about five minutes of 16 host threads for all of c2.  Prints one progress line per chunk."""
import os
import sys
import time

R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R)
import numpy as np

from allwave_amd import ffi, synth
from oracle import oracle as O

cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = synth.CONFIGS[cfgname]
if os.environ.get("CFG_SCORES"):  # another penalty set on the same read set, e.g. the CLI's ANI presets (main.rs:83-124)
    cfg = dict(cfg, scores=tuple(int(x) for x in os.environ["CFG_SCORES"].split(",")))
    cfgname += "[" + os.environ["CFG_SCORES"] + "]"
kw = {"mixed_lengths": cfg["mixed_lengths"]} if "mixed_lengths" in cfg else {}
data, offs, ids = synth.generate(cfg["nseq"], cfg["length"], cfg["d"], cfg["seed"], **kw)
if cfg["sparsify"] == "none":
    pairs = synth.all_pairs(cfg["nseq"])
else:  # the reference's sparsifiers (iterator.rs:256-334, mash.rs, knn_graph.rs) via the C++ host mirror
    from allwave_amd import host as H
    t0 = time.time()
    seqs = [bytes(data[offs[i]:offs[i + 1]]) for i in range(cfg["nseq"])]
    pairs = np.asarray(H.plan_pairs(ids, seqs, cfg["sparsify"]), dtype=np.int32).reshape(-1, 2)
    del seqs
    print("%s: -p %s keeps %d of %d pairs (planner %.1f s)" % (cfgname, cfg["sparsify"], len(pairs),
          cfg["nseq"] * (cfg["nseq"] - 1), time.time() - t0), flush=True)
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else len(pairs) - first
pairs = pairs[first:first + count]
threads = min(16, len(os.sched_getaffinity(0)))
eng = ffi.Engine()
eng.set_sequences((data, offs))
budget = float(sys.argv[4]) if len(sys.argv) > 4 else 1e9   # stop once the oracle has used this many seconds
CH = int(sys.argv[5]) if len(sys.argv) > 5 else 8192
bad = total = 0
t_gpu = t_cpu = 0.0
for a in range(0, len(pairs), CH):
    chunk = pairs[a:a + CH]
    t0 = time.time()
    res, cigs = eng.align_pairs(cfg["scores"], chunk)
    t1 = time.time()
    secs, ores, _, _ = O.all_pairs(data, offs, chunk, cfg["scores"], nthreads=threads, fast_overlap=True)
    t_gpu += t1 - t0
    t_cpu += secs
    ok = ((res["status"] == 0) & (ores["status"] == 0) & (res["penalty"] == ores["penalty"]) &
          (res["cigar_len"] == ores["cigar_len"].astype(np.uint32)) & (res["num_matches"] == ores["num_matches"]) &
          (res["num_mismatches"] == ores["num_mismatches"]) & (res["num_ins"] == ores["num_ins_text"]) &
          (res["num_del"] == ores["num_del_pattern"]))
    for i in range(len(chunk)):
        if ok[i] and O.fnv1a(cigs[i]) != int(ores["cigar_hash"][i]):
            ok[i] = False
    nb = int((~ok).sum())
    if nb:
        i = int(np.argmin(ok))
        print("MISMATCH first at pair", first + a + i, tuple(chunk[i]), "gpu", res["status"][i], res["penalty"][i],
              "oracle", ores["status"][i], ores["penalty"][i], flush=True)
    bad += nb
    total += len(chunk)
    print("%s pairs %d..%d: bad %d (cumulative %d of %d)  gpu %.1f s  oracle %.1f s on %d threads"
          % (cfgname, first + a, first + a + len(chunk) - 1, nb, bad, total, t_gpu, t_cpu, threads), flush=True)
    if t_cpu > budget:
        break
if len(sys.argv) > 6 and sys.argv[6] == "all":
    # the config's whole pair list on the GPU alone: every pair completes, consumes both sequences, and the
    # op counts add up to the op-byte length (oracle-free invariants at full size) -- and how long it takes
    ln = np.diff(offs).astype(np.int64)
    t0 = time.time()
    kms = 0.0
    nbad = ncells = 0
    for a in range(0, len(pairs), 4096):
        chunk = pairs[a:a + 4096]
        res, _ = eng.align_pairs(cfg["scores"], chunk, want_cigars=False)
        st = eng.stats()
        kms += st.kernel_ms
        ncells += int(st.cell_steps)
        good = ((res["status"] == 0) & (res["q_end"] == ln[chunk[:, 0]]) & (res["t_end"] == ln[chunk[:, 1]]) &
                (res["cigar_len"] == res["num_matches"] + res["num_mismatches"] + res["num_ins"] + res["num_del"]) &
                (res["penalty"] >= 0))
        nbad += int((~good).sum())
        print("%s gpu-only pairs %d..%d: bad %d, %.1f s so far (kernel %.1f s)" % (cfgname, a, a + len(chunk) - 1, nbad, time.time() - t0, kms * 1e-3), flush=True)
    bp = int(ln[pairs[:, 0]].sum())
    print("ALL %s: %d pairs, %d failed invariants; %.3e bp in %.1f s wall / %.1f s kernel = %.1f Mbp/s (kernel), %.3e cell-steps/s"
          % (cfgname, len(pairs), nbad, bp, time.time() - t0, kms * 1e-3, bp / (kms * 1e-3) / 1e6, ncells / (kms * 1e-3)), flush=True)
    bad += nbad
print("TOTAL %s: %d mismatches in %d pairs (penalty, length, M/X/I/D counts, FNV-1a of the op bytes)" % (cfgname, bad, total))
sys.exit(1 if bad else 0)
