#!/bin/bash
# usage (GPU box): bash scratch/r03_waits.sh <tag>   -- where the waves of config 2 wait: latency counters (LEVEL / INSTS), instruction fetch
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-w}; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_SMEM" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LEVEL_WAVES SQ_WAVES"; do
  i=$((i+1))
  rm -rf $O/waits_${T}_$i
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/waits_${T}_$i -- python3 $R/scratch/exp.py --reps 1 > $O/waits_${T}_$i.log 2>&1 || { tail -5 $O/waits_${T}_$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
ctr = collections.defaultdict(float); dur = 0
for f in glob.glob("$O/waits_${T}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "biwfa" in r["Kernel_Name"]: ctr[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(ctr): print("%-24s %.4g" % (k, ctr[k]))
g = lambda k: ctr.get(k, 0.0)
print("LDS latency (LEVEL/INSTS)  %.1f cycles" % (g("SQ_INST_LEVEL_LDS") / max(g("SQ_INSTS_LDS"), 1)))
print("VMEM latency (LEVEL/INSTS) %.1f cycles" % (g("SQ_INST_LEVEL_VMEM") / max(g("SQ_INSTS_VMEM"), 1)))
print("IFETCH latency             %.1f cycles, fetches per wave-cycle %.4f" % (g("SQ_IFETCH_LEVEL") / max(g("SQ_IFETCH"), 1), g("SQ_IFETCH") / max(g("SQ_WAVE_CYCLES"), 1)))
print("wait_any / wave_cycles     %.3f   wait_inst_any / wave_cycles %.3f   wait_inst_lds / wave_cycles %.3f" % (g("SQ_WAIT_ANY") / max(g("SQ_WAVE_CYCLES"), 1), g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1), g("SQ_WAIT_INST_LDS") / max(g("SQ_WAVE_CYCLES"), 1)))
PY
