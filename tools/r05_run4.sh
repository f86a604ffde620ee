#!/bin/bash
# round 5, GPU call 4: backward cell kernel with exact waits (WT) against the predicated one, interleaved
export TMPDIR=/tmp
O=gpurun_out/r05_run4; mkdir -p $O
for rep in 1 2 3; do
  echo "--- product (WT)"; python tools/xna_bwd_bench.py 2>&1 | grep -v amdgpu.ids
  echo "--- variant (predicated)"; NAF_HIP_LIB=tools/bin/libnaf_nowt.so python tools/xna_bwd_bench.py 2>&1 | grep -v amdgpu.ids
done | tee $O/bwd_ab.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "backward or autograd or train" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
