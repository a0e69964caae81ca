#!/bin/bash
# usage (GPU box): bash scratch/r02_campaigns.sh  -- ANI-preset parity on config 2's read set + config 5 by pair class
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
: > $O/parity_c2_ani_presets_r02.log
for sc in "0,7,12,2,36,1" "0,4,6,2,18,1" "0,3,4,1" "0,1,1,1"; do
  CFG_SCORES=$sc timeout -k 10 300 python tests/campaigns/config_full.py c2 0 3072 >> $O/parity_c2_ani_presets_r02.log 2>&1
done
grep TOTAL $O/parity_c2_ani_presets_r02.log
: > $O/c5_classes_r02.log
for sel in rows16 shorttext shortpattern; do
  timeout -k 10 300 python scratch/c45.py c5 0 0 $sel 2>&1 | head -1 >> $O/c5_classes_r02.log
done
cat $O/c5_classes_r02.log
