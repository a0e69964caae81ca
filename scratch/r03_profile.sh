#!/bin/bash
# usage (on the GPU box, via gpurun): bash scratch/r03_profile.sh <tag> [c2|c4|c5] [skip-bench]
# 1. the bench line of the config; 2. rocprofv3 kernel-trace stats of one bench step; 3. PMC passes of the same command
# (counters in passes of their own, never with a trace domain beyond --kernel-trace).
# Summarise afterwards (here): python scratch/summarize_pmc.py <tag> r03 <config>
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-v}; C=${2:-c2}; O=$R/gpurun_out; mkdir -p $O
CF=""; [ "$C" != "c2" ] && CF="--config $C"
PT=${PASS_TIMEOUT:-300}
if [ -z "$3" ]; then
  cd $R && timeout -k 10 900 python bench.py $CF $BENCH_ARGS > $O/bench_$T.json 2> $O/bench_$T.err || { tail -5 $O/bench_$T.err; exit 1; }
fi
cd /tmp; export TMPDIR=/tmp
CMD="python3 $R/bench.py $CF --steps 1 --warmup 0 --no-cpu-baseline --no-paf"
rm -rf $O/prof_$T
timeout -k 10 $PT rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -- $CMD > $O/bench_prof_$T.json 2> $O/prof_$T.err || { tail -5 $O/prof_$T.err; exit 1; }
echo "kernel-trace pass done"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_VMEM_TA_ADDR_FIFO_FULL SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rm -rf $O/pmc_${T}_$i
  timeout -k 10 $PT rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_${T}_$i -- $CMD > $O/pmc_${T}_$i.json 2> $O/pmc_${T}_$i.err || { tail -5 $O/pmc_${T}_$i.err; exit 1; }
  echo "pmc pass $i done"
done
cut -c1-600 $O/bench_prof_$T.json
