#!/bin/bash
# usage (GPU box): bash scratch/r02_ab.sh  -- A/B of engine flags / alternative builds on config 2
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/ab.log
for spec in "$@"; do
  timeout -k 10 200 python scratch/exp.py $spec >> $O/ab.log 2>&1 || { tail -5 $O/ab.log; exit 1; }
done
cat $O/ab.log
