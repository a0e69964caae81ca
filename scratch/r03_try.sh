#!/bin/bash
# usage (GPU box): bash scratch/r03_try.sh <tag> "<pytest -k expr>" "<exp.py args>" ...  -- parity subset, then A/B runs of exp.py
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-t}; K=${2:-"random_pairs or golden or config1 or config2 or edge or wide_scope or multi_step"}; O=$R/gpurun_out; mkdir -p $O
shift; shift
cd $R
if [ "$K" != "none" ]; then
  timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > $O/try_$T.pytest.log 2>&1 || { tail -30 $O/try_$T.pytest.log; exit 1; }
  tail -3 $O/try_$T.pytest.log
fi
: > $O/ab_$T.log
for spec in "$@"; do
  timeout -k 10 200 python scratch/exp.py $spec >> $O/ab_$T.log 2>&1 || { tail -5 $O/ab_$T.log; exit 1; }
done
cat $O/ab_$T.log
