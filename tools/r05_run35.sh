#!/bin/bash
# round 5, GPU call 35: what the flush's atomics cost the pitch-8 backward (variant without them: timing only, wrong dK / dV)
export TMPDIR=/tmp
O=gpurun_out/r05_run35; mkdir -p $O
for r in 1 2 3; do
  echo "== pitch 8"; BWD_BENCH_N=30 timeout 300 python tools/xna_bwd_bench.py
  echo "== pitch 8, no atomics"; NAF_HIP_LIB=$PWD/tools/bin/libnaf_p8_noat.so BWD_BENCH_N=30 timeout 300 python tools/xna_bwd_bench.py
  echo "== dense slots (previous commit)"; NAF_HIP_LIB=$PWD/tools/bin/libnaf_prev.so BWD_BENCH_N=30 timeout 300 python tools/xna_bwd_bench.py
done 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
