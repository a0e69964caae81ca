#!/bin/bash
# usage (GPU box): bash scratch/r03_final.sh <part>   -- round 3's full-size parity campaigns on the final binaries (logs -> gpurun_out/, copied to profiles/r03/)
#   a: all of config 2 (65,280 pairs) + the first 8,192 pairs of config 3        b: all of config 4 (11,741 pairs) + the first 2,048 of config 5
#   c: the CLI's other ANI presets on config 2's read set (3,072 pairs each) + two presets on config 5's first 512 pairs + the 24 k varied pairs of g8
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
case "$1" in
a)
  timeout -k 10 700 python tests/campaigns/config_full.py c2 > $O/parity_c2_full_r03.log 2>&1; tail -2 $O/parity_c2_full_r03.log
  timeout -k 10 300 python tests/campaigns/config_full.py c3 0 8192 > $O/parity_c3_first8192_r03.log 2>&1; tail -1 $O/parity_c3_first8192_r03.log ;;
b)
  timeout -k 10 700 python tests/campaigns/config_full.py c4 > $O/parity_c4_full_r03.log 2>&1; tail -2 $O/parity_c4_full_r03.log
  timeout -k 10 400 python tests/campaigns/config_full.py c5 0 2048 > $O/parity_c5_first2048_r03.log 2>&1; tail -1 $O/parity_c5_first2048_r03.log ;;
c)
  : > $O/parity_presets_r03.log
  for sc in "0,7,12,2,36,1" "0,4,6,2,18,1" "0,3,4,1" "0,1,1,1"; do
    CFG_SCORES=$sc timeout -k 10 300 python tests/campaigns/config_full.py c2 0 3072 >> $O/parity_presets_r03.log 2>&1
  done
  for sc in "0,7,12,2,36,1" "0,3,4,1"; do
    CFG_SCORES=$sc timeout -k 10 300 python tests/campaigns/config_full.py c5 0 512 >> $O/parity_presets_r03.log 2>&1
  done
  timeout -k 10 400 python tests/campaigns/g8.py >> $O/parity_presets_r03.log 2>&1
  grep -E "TOTAL|mismatch" $O/parity_presets_r03.log | tail -12 ;;
esac
