#!/bin/bash
# Round 5, end: the cell-backward fuzz (against the scalar kernel) as a campaign over three more seeds, 150 geometries each.
set -u
out=gpurun_out/r51; mkdir -p $out
for seed in 1111 2222 3333; do
  NAF_FUZZ_BWD_SEED=$seed NAF_FUZZ_BWD_CASES=150 timeout 800 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k test_cell_backward_fuzz > $out/bwd_fuzz_$seed.log 2>&1; echo "rc=$?" >> $out/bwd_fuzz_$seed.log
  grep -c "^bwd fuzz\|bwd fuzz $seed" $out/bwd_fuzz_$seed.log; tail -2 $out/bwd_fuzz_$seed.log | cut -c1-300
done
