#!/bin/bash
# round 5, GPU call 40: backward parity after the last edits, the benched backward shapes, the reference's backward protocol at four points
export TMPDIR=/tmp
O=gpurun_out/r05_run40; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "backward or bwd or autograd or train or several_runs or benched_sizes" 2>&1 | tail -4 | tee $O/pytest.txt
BWD_BENCH_N=30 timeout 300 python tools/xna_bwd_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/bwd_bench.txt
timeout 600 python tools/backward_speed_protocol.py --all 2>&1 | grep -v amdgpu.ids | tee $O/bwd_protocol.txt
