import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from allwave_amd import host as H, synth
cfg = synth.CONFIGS["c2"]
data, offs, ids = synth.generate(cfg["nseq"], cfg["length"], cfg["d"], cfg["seed"])
nsub = 96
seqs = [bytes(data[offs[i]:offs[i + 1]]) for i in range(nsub)]
for th in (16, 4, 1, 16, 32):
    t0 = time.time()
    nb, nl, secs, st = H.all_pairs_paf_count(["s%05d" % i for i in range(nsub)], seqs, "0,5,8,2,24,1", orientation="forward", device=0, format_threads=th)
    print("threads", th, "wall %.3f secs %.3f kernel %.3f d2h %.3f" % (time.time() - t0, secs, st.kernel_ms / 1e3, st.d2h_ms / 1e3), flush=True)
print(os.cpu_count(), len(os.sched_getaffinity(0)))
