#!/bin/bash
# usage (GPU box): bash scratch/r03_c5ab.sh <tag> "<lib or ->:<flags>" ...   -- config 5's pair classes (every 6th pair) under engine flags / builds
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-t}; O=$R/gpurun_out; mkdir -p $O; cd $R; shift
export C45_STRIDE=${C45_STRIDE:-6}
: > $O/c5ab_$T.log
for spec in "$@"; do
  lib=${spec%%:*}; fl=${spec##*:}
  for sel in shortpattern rows32 rows16; do
    if [ "$lib" != "-" ]; then export AWV_HIP_LIB=$R/$lib; else unset AWV_HIP_LIB; fi
    echo "== $spec $sel" >> $O/c5ab_$T.log
    timeout -k 10 300 python scratch/c45.py c5 4 $fl $sel > $O/c5ab_one.log 2>&1 || { tail -5 $O/c5ab_one.log; exit 1; }
    head -1 $O/c5ab_one.log >> $O/c5ab_$T.log
  done
done
python - <<PY
import json
cur=None
for l in open("$O/c5ab_$T.log"):
    if l.startswith("=="): cur=l.strip()
    elif l.startswith("{"):
        d=json.loads(l); print("%-60s %9.1f ms  pairs %5d  multi %.3f restarts %d" % (cur, d["kernel_ms"], d["pairs"], d["multi_frac"], d["restarts"]))
PY
