#!/bin/bash
# round 5, GPU call 9: flakiness screen of the concurrency tests (20 repetitions)
export TMPDIR=/tmp
O=gpurun_out/r05_run9; mkdir -p $O
for i in $(seq 20); do
  timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_host_threads or error_after_the_fork or two_streams_does_not_share or captured or graph" 2>&1 | tail -1
done | tee $O/flaky.txt
