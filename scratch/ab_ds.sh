#!/bin/bash
# A/B of the direction-split throughput flavour (liballwave_hip_ds.so, -DAWV_THRU_WG=128) against the
# one-wave flavour on all of config 2, interleaved runs in one gpurun call; parity of the variant first.
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
AWV_HIP_LIB=$PWD/allwave_amd/liballwave_hip_ds.so timeout -k 10 200 python tests/campaigns/config_full.py c2 0 4096 1000 4096
for i in 1 2; do
  for v in hip hip_ds; do
    echo "== $v"
    AWV_HIP_LIB=$PWD/allwave_amd/liballwave_$v.so timeout -k 10 120 python scratch/g4.py 4096 65280
  done
done
