#!/bin/bash
# round 5, GPU call 34: backward v2 at 7 x 7 with a column pitch of 8 slots (the flush of the leaving column = one aligned half of a key tile, asm atomics in saddr form)
export TMPDIR=/tmp
O=gpurun_out/r05_run34; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "backward or bwd or autograd or train" 2>&1 | tail -5 | tee $O/pytest.txt
for r in 1 2 3; do
  echo "== pitch 8 (new)"; BWD_BENCH_N=30 timeout 300 python tools/xna_bwd_bench.py
  echo "== dense slots (previous commit)"; NAF_HIP_LIB=$PWD/tools/bin/libnaf_prev.so BWD_BENCH_N=30 timeout 300 python tools/xna_bwd_bench.py
done 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
