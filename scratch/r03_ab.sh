#!/bin/bash
# usage (GPU box): bash scratch/r03_ab.sh <tag> <parity-lib|none> "<exp.py args>" ...
# parity subset of the GPU suite on an alternative build (AWV_HIP_LIB), then same-box A/B runs of scratch/exp.py
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-t}; PL=${2:-none}; O=$R/gpurun_out; mkdir -p $O
shift; shift
cd $R
if [ "$PL" != "none" ]; then
  AWV_HIP_LIB=$R/$PL timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_pairs or golden or config1 or config2 or edge or wide_scope or multi_step or gotoh or wide16" > $O/try_$T.pytest.log 2>&1 || { tail -30 $O/try_$T.pytest.log; exit 1; }
  tail -2 $O/try_$T.pytest.log
fi
: > $O/ab_$T.log
for spec in "$@"; do
  timeout -k 10 200 python scratch/exp.py $spec >> $O/ab_$T.log 2>&1 || { tail -5 $O/ab_$T.log; exit 1; }
done
python - <<PY
import json
for l in open("$O/ab_$T.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("%-10s rep %d  %8.2f ms  %6.1f Mbp/s  cells %d  multi %.4f deep %.4f  restarts %d  win_single %d win_multi %d  pen_sum %d bad %d  clk %.3f" % (
            d["tag"], d["rep"], d["kernel_ms"], d["Mbp_s"], d["cells"], d["multi_frac"], d.get("deep_frac", 0), d["restarts"], d["win_single"], d["win_multi"], d["pen_sum"], d["bad"], d.get("clock_ghz", 0)))
PY
