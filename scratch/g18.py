import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from allwave_amd import ffi, synth
data, offs, ids = synth.generate(256, 10000, 0.05, 2)
pairs = synth.all_pairs(256)
n = 32768
def run(e):
    ts = []
    for k in range(2):
        res,_ = e.align_pairs((0,5,8,2,24,1), pairs[:n], want_cigars=False)
        ts.append(e.stats().kernel_ms)
    return " ".join("%.0f" % t for t in ts)
engines = []
for rep in range(5):   # engines stay alive: every arena is a fresh allocation
    e = ffi.Engine(workgroups=4096, flags=ffi.AWV_F_KEEP_ON_DEVICE | ffi.AWV_F_ONE_WAVE)
    e.set_sequences((data, offs))
    print("alive", rep, "kernel_ms", run(e), flush=True)
    engines.append(e)
print("re-run of each while all alive:", [run(e) for e in engines], flush=True)
for e in engines: e.close()
