#!/bin/bash
# usage (GPU box): bash scratch/r02_final.sh <tag>  -- the round's evidence run: GPU test suite, profile set, 2-rank rehearsal, configs 4 and 5
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-f}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gputest_$T.log 2>&1; tail -3 $O/gputest_$T.log
bash scratch/r02_profile.sh $T || exit 1
cd $R
timeout -k 10 300 python bench.py --gpus 2 --rehearse-one-gpu --steps 1 --warmup 0 > $O/bench_2rank_$T.json 2> $O/bench_2rank_$T.err; cut -c1-700 $O/bench_2rank_$T.json
timeout -k 10 300 python scratch/c45.py c4 12 > $O/c4_$T.log 2>&1; cat $O/c4_$T.log
timeout -k 10 500 python scratch/c45.py c5 48 > $O/c5_$T.log 2>&1; cat $O/c5_$T.log
timeout -k 10 400 python tests/campaigns/config_full.py c3 0 8192 > $O/parity_c3_first8192_$T.log 2>&1; tail -1 $O/parity_c3_first8192_$T.log
timeout -k 10 120 python scratch/exp.py --pairs 256 --reps 3 --tag small256 > $O/small256_$T.log 2>&1; tail -1 $O/small256_$T.log | cut -c1-200
