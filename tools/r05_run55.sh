#!/bin/bash
# Round 5, end: the empty last key tile of 13 x 13 / 15 x 15 skipped in the eight-wave backward -- parity (oracle, fuzz against the scalar kernel, multi-run, G2-k15 at
# full size), then timings (to compare with gpurun r53 / r54: G2-k15 1.45-1.53 ms, 13 x 13 at C = 1024 1.14-1.16 ms).
set -u
out=gpurun_out/r55; mkdir -p $out
NAF_FUZZ_BWD_SEED=5555 NAF_FUZZ_BWD_CASES=120 timeout 600 python -m pytest tests -m gpu -q -s -k "(test_xna_backward_matches_oracle and (13 or 15)) or test_cell_backward_fuzz or test_cell_backward_walks or (benched_sizes and k15)" > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
tail -3 $out/tests.log | cut -c1-300
for i in 1 2 3; do python tools/bwd_k15_time.py --fast 2>/dev/null | grep "k 1[35]"; done > $out/times.txt; cat $out/times.txt
