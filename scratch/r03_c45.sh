#!/bin/bash
# usage (GPU box): bash scratch/r03_c45.sh <tag> [flags]  -- configs 4 and 5 at full size (kernel time, invariants, oracle sample), VALU issue rates
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-t}; F=${2:-0}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 300 python scratch/c45.py c4 32 $F > $O/c4_$T.log 2>&1 || { tail -5 $O/c4_$T.log; exit 1; }
cat $O/c4_$T.log
timeout -k 10 500 python scratch/c45.py c5 48 $F > $O/c5_$T.log 2>&1 || { tail -5 $O/c5_$T.log; exit 1; }
cat $O/c5_$T.log
