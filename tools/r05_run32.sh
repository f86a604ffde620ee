#!/bin/bash
# round 5, GPU call 32: how many V key tiles the query waves of the wave-specialised backward keep in registers (96 = the product, 72, 48 registers of them)
export TMPDIR=/tmp
O=gpurun_out/r05_run32; mkdir -p $O
for r in 1 2 3; do
  echo "== 96 registers of V fragments (product)"; BWD_BENCH_N=30 timeout 300 python tools/xna_bwd_bench.py
  echo "== 72"; NAF_HIP_LIB=tools/bin/libnaf_v72.so BWD_BENCH_N=30 timeout 300 python tools/xna_bwd_bench.py
  echo "== 48"; NAF_HIP_LIB=tools/bin/libnaf_v48.so BWD_BENCH_N=30 timeout 300 python tools/xna_bwd_bench.py
done 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
