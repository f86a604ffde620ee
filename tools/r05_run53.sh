#!/bin/bash
# Round 5, end: 15 x 15 in chunks of 64 on the eight-wave backward (P / dS rows of 240 slots) against chunks of 32 (NAF_BWD_K15_C32=1): parity first, then A/B.
set -u
out=gpurun_out/r53; mkdir -p $out
NAF_FUZZ_BWD_SEED=4444 NAF_FUZZ_BWD_CASES=120 timeout 600 python -m pytest tests -m gpu -q -s -k "(test_xna_backward_matches_oracle and 15) or test_cell_backward_fuzz or test_cell_backward_walks or (benched_sizes and k15)" > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
grep -c "bwd fuzz 4444: k 15" $out/tests.log; tail -4 $out/tests.log | cut -c1-300
for i in 1 2 3; do
  python tools/bwd_k15_time.py --fast 2>/dev/null | grep "k 15" | sed 's/^/chunks of 64  /'
  NAF_HIP_KNOBS=1 NAF_BWD_K15_C32=1 python tools/bwd_k15_time.py --fast 2>/dev/null | grep "k 15" | sed 's/^/chunks of 32  /'
done > $out/ab_k15.txt
cat $out/ab_k15.txt
