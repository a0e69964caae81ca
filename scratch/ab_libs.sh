#!/bin/bash
# usage: bash scratch/ab_libs.sh <rounds> <variant>...   -- interleaved config-2 kernel times of
# allwave_amd/liballwave_<variant>.so builds in one gpurun call (the first is the baseline)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
n=$1; shift
for i in $(seq $n); do
  for v in "$@"; do
    echo "== $v"
    AWV_HIP_LIB=$PWD/allwave_amd/liballwave_$v.so timeout -k 10 120 python scratch/g4.py 4096 65280 || exit 1
  done
done
