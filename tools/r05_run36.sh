#!/bin/bash
# round 5, GPU call 36: backward v2 with two sets of row fragments (requests two rounds ahead) against one set (NAF_BWD2_ONE=1)
export TMPDIR=/tmp
O=gpurun_out/r05_run36; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "backward or bwd or autograd or train" 2>&1 | tail -5 | tee $O/pytest.txt
for r in 1 2 3; do
  echo "== two sets (new)"; BWD_BENCH_N=30 timeout 300 python tools/xna_bwd_bench.py
  echo "== one set"; NAF_HIP_KNOBS=1 NAF_BWD2_ONE=1 BWD_BENCH_N=30 timeout 300 python tools/xna_bwd_bench.py
done 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
