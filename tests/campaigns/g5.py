"""Throughput + spot parity on config-4- and config-5-shaped inputs (reduced sequence counts)."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from allwave_amd import ffi, synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "oracle"))
import oracle

def run(name, nseq, length, d, seed, mixed, npairs_check, max_pairs=None):
    data, offs, ids = synth.generate(nseq, length, d, seed, mixed_lengths=mixed)
    pairs = synth.all_pairs(nseq)
    if max_pairs: pairs = pairs[:max_pairs]
    scores = (0, 5, 8, 2, 24, 1)
    e = ffi.Engine(flags=ffi.AWV_F_KEEP_ON_DEVICE)
    e.set_sequences((data, offs))
    t0 = time.time()
    res, _ = e.align_pairs(scores, pairs, want_cigars=False)
    wall = time.time() - t0
    st = e.stats()
    bp = int(sum(int(offs[a + 1] - offs[a]) for a, b in pairs))
    print("%s: pairs %d launches %d kernel_ms %.1f wall %.2f s  pairs/s %.1f  Mbp/s %.2f  cells/s %.3e  bad_status %d scratch GiB %.1f"
          % (name, len(pairs), st.launches, st.kernel_ms, wall, len(pairs) / (st.kernel_ms * 1e-3), bp / (st.kernel_ms * 1e-3) / 1e6,
             st.cell_steps / (st.kernel_ms * 1e-3), int((res["status"] != 0).sum()), st.scratch_bytes / 2**30), flush=True)
    e.close()
    # spot parity with CIGARs
    e = ffi.Engine()
    e.set_sequences((data, offs))
    sel = pairs[:: max(1, len(pairs) // npairs_check)][:npairs_check]
    res, cigs = e.align_pairs(scores, sel)
    al = oracle.Aligner(scores)
    bad = 0
    t0 = time.time()
    for i, (a, b) in enumerate(sel):
        pen, ops = al.align(bytes(data[offs[a]:offs[a + 1]]), bytes(data[offs[b]:offs[b + 1]]))
        if res["status"][i] != 0 or res["penalty"][i] != pen or cigs[i] != ops: bad += 1
    print("  spot parity: %d pairs, bad %d (oracle %.1f s)" % (len(sel), bad, time.time() - t0), flush=True)
    e.close()

which = sys.argv[1:] or ["c4", "c5"]
n4 = int(os.environ.get("G5_N4", "24"))
if "c4" in which: run("c4-shaped %dx100kbp 2%%" % n4, n4, 100000, 0.02, 4, None, 2)
if "c5" in which: run("c5-shaped 128 mixed 1-50kbp 10%", 128, 50000, 0.10, 5, (1000, 50000), 6, max_pairs=4096)
