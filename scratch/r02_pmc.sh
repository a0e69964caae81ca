#!/bin/bash
# usage (GPU box): bash scratch/r02_pmc.sh <tag> [exp.py args]  -- PMC passes of one config-2 launch (scratch/exp.py --reps 1)
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-p}; shift; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rm -rf $O/pmc_${T}_$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_${T}_$i -- python3 $R/scratch/exp.py --reps 1 "$@" > $O/pmc_${T}_$i.log 2>&1 || { tail -5 $O/pmc_${T}_$i.log; exit 1; }
  echo "pmc pass $i done: $(tail -1 $O/pmc_${T}_$i.log | cut -c1-160)"
done
python3 $R/scratch/pmcsum.py $O/pmc_${T}_1 $O/pmc_${T}_2 $O/pmc_${T}_3 $O/pmc_${T}_4 | tee $O/pmc_${T}.txt
