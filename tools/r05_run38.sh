#!/bin/bash
# round 5, GPU call 37-38: backward v2 at 9 x 9 with Dv = 192 (88-slot P / dS rows) and Dv = 256 (one P / dS buffer) -- parity, A/B against the previous library (four-wave kernel at that shape)
export TMPDIR=/tmp
O=gpurun_out/r05_run38; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "backward or bwd or autograd or train" 2>&1 | tail -5 | tee $O/pytest.txt
for r in 1 2 3; do
  echo "== new"; BWD_BENCH_N=30 timeout 300 python tools/xna_bwd_bench.py
  echo "== previous commit"; NAF_HIP_LIB=$PWD/tools/bin/libnaf_prev.so BWD_BENCH_N=30 timeout 300 python tools/xna_bwd_bench.py
done 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
