#!/bin/bash
# usage (GPU box): bash scratch/r02_final2.sh <tag>  -- second half of the round's evidence run: bench lines of configs 4 and 5,
# the 16-bit min(h, v) rows against the 32-bit-row kernels, counter passes of configs 4 and 5
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-g}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 300 python bench.py --config c4 --steps 2 --warmup 1 > $O/bench_c4_$T.json 2> $O/bench_c4_$T.err; cut -c1-600 $O/bench_c4_$T.json
timeout -k 10 500 python bench.py --config c5 --steps 1 --warmup 0 > $O/bench_c5_$T.json 2> $O/bench_c5_$T.err; cut -c1-600 $O/bench_c5_$T.json
timeout -k 10 300 python scratch/wide16_check.py 16 24 > $O/wide16_$T.log 2>&1; tail -4 $O/wide16_$T.log
timeout -k 10 200 python scratch/t32_check.py 1024 16 > $O/t32_$T.log 2>&1; timeout -k 10 200 python scratch/t32_check.py 3000 >> $O/t32_$T.log 2>&1; cat $O/t32_$T.log | cut -c1-200
bash scratch/r02_pmc_c45.sh c4$T c4 all 1 > $O/pmc_c4_$T.out 2>&1; tail -30 $O/pmc_c4_$T.out | head -4
PASSES="1 2 4" bash scratch/r02_pmc_c45.sh c5$T c5 all 1 > $O/pmc_c5_$T.out 2>&1; grep -a "kernel" $O/pmc_c5_$T.out | head -6
