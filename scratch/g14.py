import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from allwave_amd import ffi, synth
data, offs, ids = synth.generate(256, 10000, 0.05, 2)
pairs = synth.all_pairs(256)
for n in (1024, 2048, 3072, 4096, 6144, 8192, 16384):
    out = []
    for name, fl in (("one", ffi.AWV_F_ONE_WAVE), ("four", ffi.AWV_F_FOUR_WAVES)):
        e = ffi.Engine(flags=fl | ffi.AWV_F_KEEP_ON_DEVICE)
        e.set_sequences((data, offs))
        res, _ = e.align_pairs((0, 5, 8, 2, 24, 1), pairs[:n], want_cigars=False)
        res, _ = e.align_pairs((0, 5, 8, 2, 24, 1), pairs[:n], want_cigars=False)
        out.append("%s %.1f ms" % (name, e.stats().kernel_ms))
        e.close()
    print(n, " | ".join(out), flush=True)
