#!/bin/bash
# Round 5, end: 11 x 11 beyond Dv = 128 as channel chunks on the eight-wave backward -- parity (oracle, multi-run, fuzz, G2-k11 at full size against the
# scalar kernel) and the interleaved A/B against the four-wave kernel on the whole head (NAF_BWD_CHUNK11=0).
set -u
out=gpurun_out/r45; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -k "(test_xna_backward_matches_oracle and 11) or test_cell_backward_fuzz or test_cell_backward_walks or (benched_sizes and k11)" > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
tail -6 $out/tests.log | cut -c1-300
for i in 1 2 3; do
  python tools/bwd_k15_time.py --fast 2>/dev/null | grep "k 11" | sed 's/^/chunks  /'
  NAF_HIP_KNOBS=1 NAF_BWD_CHUNK11=0 python tools/bwd_k15_time.py --fast 2>/dev/null | grep "k 11" | sed 's/^/whole   /'
done > $out/ab_k11.txt
cat $out/ab_k11.txt
