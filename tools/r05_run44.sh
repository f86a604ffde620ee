#!/bin/bash
# Round 5, end: the whole GPU suite on the final tree, the forward fuzz with return_weights on a quarter of the cases, the default bench line.
set -u
out=gpurun_out/r44; mkdir -p $out
timeout 1300 python -m pytest tests -m gpu -x -q --durations=12 > $out/gputest.log 2>&1; echo "rc=$?" >> $out/gputest.log
tail -18 $out/gputest.log | cut -c1-200
NAF_FUZZ_CASES=200 timeout 600 python -m pytest tests/test_gpu_fuzz_forward.py -m gpu -q -s > $out/fuzz_forward_scores.log 2>&1; echo "rc=$?" >> $out/fuzz_forward_scores.log
tail -3 $out/fuzz_forward_scores.log | cut -c1-300
python bench.py > $out/bench.json 2> $out/bench.err; tail -c 400 $out/bench.json
