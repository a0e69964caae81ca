#!/bin/bash
# usage (GPU box): bash scratch/r02_pmc_c45.sh <tag> <c4|c5> <select> <stride>  -- PMC passes of a config-4/5 pair class
set -o pipefail
R=$GRAFT_REPO_ROOT; T=$1; C=$2; S=$3; export C45_STRIDE=${4:-1}; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
PASSES=${PASSES:-1 2 3 4}
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVES"; do
  i=$((i+1))
  case " $PASSES " in *" $i "*) ;; *) continue;; esac
  rm -rf $O/pmc_${T}_$i
  timeout -k 10 ${PASS_TIMEOUT:-400} rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_${T}_$i -- python3 $R/scratch/c45.py $C 2 0 $S > $O/pmc_${T}_$i.log 2>&1 || { tail -5 $O/pmc_${T}_$i.log; exit 1; }
  echo "pmc pass $i done: $(head -1 $O/pmc_${T}_$i.log | cut -c1-400)"
done
python3 $R/scratch/pmc_c45_sum.py $T $O
