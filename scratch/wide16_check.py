"""AWV_WIDE16 rows (16-bit min(h, v) rows for pairs whose LONGER sequence has 32760 bases or more) against the 32-bit-row
kernels on the same pairs: every result field and every CIGAR, plus an oracle sample.
usage: python scratch/wide16_check.py [stride] [oracle_pairs]"""
import os, sys, json, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from allwave_amd import ffi, synth, host as H
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nor = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfg = synth.CONFIGS["c5"]
data, offs, ids = synth.generate(cfg["nseq"], cfg["length"], cfg["d"], cfg["seed"], mixed_lengths=cfg["mixed_lengths"])
lens = (offs[1:] - offs[:-1]).astype(np.int64)
seqs = [bytes(data[offs[i]:offs[i + 1]]) for i in range(cfg["nseq"])]
pairs = np.asarray(H.plan_pairs(ids, seqs, cfg["sparsify"]), dtype=np.int32).reshape(-1, 2)
del seqs
ql, tl = lens[pairs[:, 0]], lens[pairs[:, 1]]
sel = (np.maximum(ql, tl) >= 32760) & (np.minimum(ql, tl) < 32760)
pairs = np.ascontiguousarray(pairs[sel][::stride])
print("pairs", len(pairs), "short text", int((lens[pairs[:, 1]] < 32760).sum()), "short pattern", int((lens[pairs[:, 0]] < 32760).sum()), flush=True)
out = {}
for name, flags in (("wide16", 0), ("rows32", ffi.AWV_F_NO_WIDE16)):
    e = ffi.Engine(flags=flags)
    e.set_sequences((data, offs))
    res, cig = e.align_pairs(cfg["scores"], pairs)
    st = e.stats()
    e.close()
    out[name] = (res, [zlib.crc32(bytes(c)) for c in cig])
    print(json.dumps({"rows": name, "kernel_ms": round(st.kernel_ms, 1), "launches": st.launches, "failed": int((res["status"] != 0).sum()),
                      "cell_steps": st.cell_steps, "multi_frac": round(st.multi_cell_steps / max(1, st.cell_steps), 4), "restarts": st.restarts}), flush=True)
a, b = out["wide16"], out["rows32"]
bad = 0
for f in a[0].dtype.names:
    n = int((a[0][f] != b[0][f]).sum())
    if n: print("field", f, "differs on", n, "pairs"); bad += n
nc = sum(1 for x, y in zip(a[1], b[1]) if x != y)
print(json.dumps({"pairs": len(pairs), "field_mismatches": bad, "cigar_mismatches": nc}), flush=True)
if nor:
    from oracle import oracle as O
    idx = np.unique(np.linspace(0, len(pairs) - 1, nor).astype(np.int64))
    sub = np.ascontiguousarray(pairs[idx])
    e = ffi.Engine()
    e.set_sequences((data, offs))
    gres, gc = e.align_pairs(cfg["scores"], sub)
    e.close()
    secs, ores, _, _ = O.all_pairs(data, offs, sub, cfg["scores"], nthreads=min(16, len(os.sched_getaffinity(0))), fast_overlap=True)
    ob = sum(1 for i in range(len(sub)) if gres["status"][i] != 0 or gres["penalty"][i] != ores["penalty"][i] or O.fnv1a(gc[i]) != int(ores["cigar_hash"][i]))
    print(json.dumps({"oracle_sample": len(sub), "mismatches": ob, "oracle_s": round(secs, 1)}), flush=True)
sys.exit(1 if (bad or nc) else 0)
