#!/bin/bash
# Round 5, end: 13 x 13 / 15 x 15 as chunks of 64 / 32 on the EIGHT-wave backward (NAF_BWD_BIG8=1) against chunks of 128 / 64 on the four-wave kernel:
# parity of the variant (oracle cases, cell fuzz against the scalar kernel, G2-k15 at full size) and an interleaved A/B.
set -u
out=gpurun_out/r47; mkdir -p $out
NAF_HIP_KNOBS=1 NAF_BWD_BIG8=1 timeout 900 python -m pytest tests -m gpu -q -k "(test_xna_backward_matches_oracle and (13 or 15)) or test_cell_backward_fuzz or (benched_sizes and k15)" > $out/tests_big8.log 2>&1; echo "rc=$?" >> $out/tests_big8.log
tail -6 $out/tests_big8.log | cut -c1-300
for i in 1 2 3; do
  python tools/bwd_k15_time.py --fast 2>/dev/null | grep "k 1[35]" | sed 's/^/four-wave   /'
  NAF_HIP_KNOBS=1 NAF_BWD_BIG8=1 python tools/bwd_k15_time.py --fast 2>/dev/null | grep "k 1[35]" | sed 's/^/eight-wave  /'
done > $out/ab_big8.txt
cat $out/ab_big8.txt
