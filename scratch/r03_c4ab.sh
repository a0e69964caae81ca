#!/bin/bash
# usage (GPU box): bash scratch/r03_c4ab.sh <tag> <libA|-> <libB|-> ...   -- config 4 at full size, each build twice, interleaved
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-t}; O=$R/gpurun_out; mkdir -p $O; cd $R; shift
: > $O/c4ab_$T.log
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" != "-" ]; then export AWV_HIP_LIB=$R/$lib; else unset AWV_HIP_LIB; fi
    timeout -k 10 300 python scratch/c45.py c4 4 0 > $O/c4ab_one.log 2>&1 || { tail -5 $O/c4ab_one.log; exit 1; }
    echo "$lib $(head -1 $O/c4ab_one.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["kernel_ms"], d["restarts"], d["failed_invariants"])')" >> $O/c4ab_$T.log
  done
done
cat $O/c4ab_$T.log
