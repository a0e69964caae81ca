#!/bin/bash
# usage (GPU box): bash scratch/r02_cal.sh  -- VALU issue rates, FETCH/WRITE_SIZE calibration, baseline bench
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 120 $R/scratch/bin/valurate > $O/valurate.jsonl || exit 1
echo "valurate done"
timeout -k 10 120 $R/scratch/bin/fetchcal > $O/fetchcal.jsonl || exit 1
echo "fetchcal done"
rm -rf $O/cal_fetch $O/cal_write
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -- $R/scratch/bin/fetchcal > $O/cal_fetch.log 2>&1 || { tail -5 $O/cal_fetch.log; exit 1; }
echo "pmc FETCH_SIZE done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/cal_write -- $R/scratch/bin/fetchcal > $O/cal_write.log 2>&1 || { tail -5 $O/cal_write.log; exit 1; }
echo "pmc WRITE_SIZE done"
cd $R && timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench_r02_base.json 2> $O/bench_r02_base.err || { tail -5 $O/bench_r02_base.err; exit 1; }
cut -c1-600 $O/bench_r02_base.json
