#!/bin/bash
# The command-line driver on config 3's read set (4096 x 10 kbp) with -p random:0.03 (~500 k pairs, ~10 GB of
# CIGAR op bytes => two 8 GiB launch batches): the multi-batch path at real arena sizes, helper-thread sink included.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'PY'
from allwave_amd import synth
cfg = synth.CONFIGS["c3"]
data, offs, ids = synth.generate(cfg["nseq"], cfg["length"], cfg["d"], cfg["seed"])
with open("/tmp/c3.fa", "wb") as f:
    for i, name in enumerate(ids):
        f.write(b">" + name.encode() + b"\n" + bytes(data[offs[i]:offs[i + 1]]) + b"\n")
PY
s=$(date +%s.%N)
AWV_TIMING=1 timeout -k 10 500 ./allwave_amd/allwave_hip -i /tmp/c3.fa -o /tmp/c3.paf -p random:0.03 -s 0,5,8,2,24,1 -t 16 --forward-only 2> /tmp/c3.err || { tail -5 /tmp/c3.err; exit 1; }
e=$(date +%s.%N)
grep -E "kernel done|sink|cigars on host|arenas" /tmp/c3.err | cut -c1-120
tail -1 /tmp/c3.err
python - <<PY
import re, random
t = $e - $s
n = 0; bad = 0; seen = set()
with open("/tmp/c3.paf") as f:
    for line in f:
        n += 1
        if n % 97: continue
        F = line.rstrip("\n").split("\t")
        ql, tl = int(F[1]), int(F[6])
        cg = [x for x in F if x.startswith("cg:Z:")][0][5:]
        q = r = 0
        for c, op in re.findall(r"(\d+)([=XID])", cg):
            c = int(c)
            if op in "=X": q += c; r += c
            elif op == "I": q += c      # standard CIGAR in the PAF: I consumes the query
            else: r += c
        if (q, r) != (ql, tl) or int(F[3]) != ql or int(F[8]) != tl: bad += 1
print("cli c3 -p random:0.03: %d PAF lines in %.1f s wall = %.0f lines/s (%.1f Mbp/s); %d of %d sampled lines fail full consumption" % (n, t, n / t, n * 1e4 / t / 1e6, bad, n // 97))
PY
rm -f /tmp/c3.paf
