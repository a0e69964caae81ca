#!/bin/bash
# usage: pmc2.sh <lib> <workgroups> <pairs> <outdir-prefix> : instruction-fetch / issue-stall counters
R=$GRAFT_REPO_ROOT; export AWV_HIP_LIB=$1; cd /tmp; export TMPDIR=/tmp
run() { timeout -k 10 200 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $R/gpurun_out/$4_$1 -- python3 $R/scratch/g4.py $3 $5 > $R/gpurun_out/$4_$1.log 2>&1; tail -1 $R/gpurun_out/$4_$1.log | cut -c1-100; }
run e "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" $2 $4 $3
run f "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL" $2 $4 $3
run g "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU" $2 $4 $3
