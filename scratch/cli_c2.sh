#!/bin/bash
# The command-line driver end to end on all of config 2: FASTA in -> PAF file out (the "PAF lines/sec"
# half of BASELINE.json's metric through the reference's own CLI surface).  Usage: bash scratch/cli_c2.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'PY'
from allwave_amd import synth
cfg = synth.CONFIGS["c2"]
data, offs, ids = synth.generate(cfg["nseq"], cfg["length"], cfg["d"], cfg["seed"])
with open("/tmp/c2.fa", "wb") as f:
    for i, name in enumerate(ids):
        f.write(b">" + name.encode() + b"\n" + bytes(data[offs[i]:offs[i + 1]]) + b"\n")
PY
for mode in "--forward-only" ""; do
  for rep in 1 2; do
    s=$(date +%s.%N)
    timeout -k 10 200 ./allwave_amd/allwave_hip -i /tmp/c2.fa -o /tmp/c2.paf -p none -s 0,5,8,2,24,1 -t 16 $mode 2> /tmp/c2.err || { tail -3 /tmp/c2.err; exit 1; }
    e=$(date +%s.%N)
    n=$(wc -l < /tmp/c2.paf); b=$(stat -c %s /tmp/c2.paf)
    [ "$mode" = "--forward-only" ] && cp /tmp/c2.paf /tmp/c2_fwd.paf
    python -c "t=$e-$s; print('cli c2 ${mode:-mash-orientation(default)} run $rep: %d PAF lines, %.1f MB in %.2f s wall (process start to exit) = %.0f lines/s, %.1f Mbp/s' % ($n, $b/1e6, t, $n/t, $n*10000/t/1e6))"
    if [ -n "$AWH_TIMING" ]; then cat /tmp/c2.err; else tail -1 /tmp/c2.err; fi
  done
done
head -c 300 /tmp/c2.paf; echo
