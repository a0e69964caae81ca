#!/bin/bash
# usage (GPU box): bash scratch/r02_try.sh <tag> [pytest -k expr]  -- quick parity subset + bench (new path vs single-step)
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-t}; K=${2:-"random_pairs or golden or config1 or config2 or row_width_and or wide_scope"}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > $O/try_$T.pytest.log 2>&1 || { tail -30 $O/try_$T.pytest.log; exit 1; }
tail -3 $O/try_$T.pytest.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-paf --steps 2 --warmup 1 > $O/try_$T.bench.json 2> $O/try_$T.bench.err || { tail -5 $O/try_$T.bench.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/try_$T.bench.json"))
print("value %.1f Mbp/s  ms/step %.1f  kernel %.1f ms  frac %.3f" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
PY
