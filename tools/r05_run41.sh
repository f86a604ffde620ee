#!/bin/bash
# Round 5, late: (1) the backward tests with a TRUE scalar-kernel reference (naf_xna_bwd_args.path) and the channel-chunked 13 x 13 / 15 x 15
# cell backward; (2) whole-forward fuzz against the oracle; (3) timings of the large windows' backward.
set -u
out=gpurun_out/r41; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q -k "backward" --durations=8 > $out/backward_tests.log 2>&1; echo "rc=$?" >> $out/backward_tests.log
tail -25 $out/backward_tests.log
NAF_FUZZ_CASES=${NAF_FUZZ_CASES:-200} timeout 900 python -m pytest tests/test_gpu_fuzz_forward.py -m gpu -q -s > $out/fuzz_forward.log 2>&1; echo "rc=$?" >> $out/fuzz_forward.log
grep -c "^fuzz" $out/fuzz_forward.log; tail -4 $out/fuzz_forward.log
timeout 300 python tools/bwd_k15_time.py > $out/bwd_k15_time.txt 2>&1; cat $out/bwd_k15_time.txt
