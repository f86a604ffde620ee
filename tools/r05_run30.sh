#!/bin/bash
# round 5, GPU call 30: wave-specialised backward at Dv = 256 (7 x 7: three of four V key tiles resident) -- parity, interleaved A/B against the four-wave kernel
export TMPDIR=/tmp
O=gpurun_out/r05_run30; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "backward or bwd or autograd or train" 2>&1 | tail -5 | tee $O/pytest.txt
for r in 1 2 3; do
  echo "== eight waves (product)"; BWD_BENCH_N=30 timeout 300 python tools/xna_bwd_bench.py
  echo "== four waves (NAF_BWD_V1=1)"; NAF_HIP_KNOBS=1 NAF_BWD_V1=1 BWD_BENCH_N=30 timeout 300 python tools/xna_bwd_bench.py
done 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
