#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for lib in "" scratch/bin/lib_m256_8.so scratch/bin/lib_m512_8.so scratch/bin/lib_m512_6.so; do
  AWV_HIP_LIB=${lib:+$R/$lib} timeout -k 10 200 python scratch/c45.py c4 0 2>&1 | head -1 | cut -c1-330
done
