#!/bin/bash
# round 3, first look: where the step-by-step window-steps come from (diag build), the cycle-stamp profile, the baseline time
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 200 python scratch/diag.py scratch/bin/liballwave_hip_diag.so 16384 > $O/r03_diag.log 2>&1 || { tail -5 $O/r03_diag.log; exit 1; }
cat $O/r03_diag.log
timeout -k 10 200 python scratch/prof.py scratch/bin/liballwave_hip_prof.so 0 16384 > $O/r03_prof.log 2>&1 || { tail -5 $O/r03_prof.log; exit 1; }
cat $O/r03_prof.log
timeout -k 10 200 python scratch/exp.py --tag base --reps 2 > $O/r03_base.log 2>&1 || { tail -5 $O/r03_base.log; exit 1; }
cat $O/r03_base.log
