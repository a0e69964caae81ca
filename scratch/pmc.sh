#!/bin/bash
# usage: pmc.sh <lib> <workgroups> <pairs> <outdir-prefix>
R=$GRAFT_REPO_ROOT; export AWV_HIP_LIB=$1; cd /tmp; export TMPDIR=/tmp
run() { timeout -k 10 200 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $R/gpurun_out/$4_$1 -- python3 $R/scratch/g4.py $3 $5 > $R/gpurun_out/$4_$1.log 2>&1; tail -1 $R/gpurun_out/$4_$1.log | cut -c1-100; }
run a "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" $2 $4 $3
run b "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" $2 $4 $3
run c "FETCH_SIZE" $2 $4 $3
run d "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" $2 $4 $3
