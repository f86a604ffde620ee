#!/bin/bash
# round 5, GPU call 39: backward v2 after the register work for the wide 9 x 9 shapes -- A/B against the library before it, all benched shapes
export TMPDIR=/tmp
O=gpurun_out/r05_run39; mkdir -p $O
for r in 1 2 3; do
  echo "== new"; BWD_BENCH_N=30 timeout 300 python tools/xna_bwd_bench.py
  echo "== before"; NAF_HIP_LIB=$PWD/tools/bin/libnaf_prev.so BWD_BENCH_N=30 timeout 300 python tools/xna_bwd_bench.py
done 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
