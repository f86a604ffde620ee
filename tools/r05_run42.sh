#!/bin/bash
# Round 5, late: the seven one-pixel-cell outliers of the first fuzz campaign through the decomposed check; the chunked 13 x 13 / 15 x 15 backward
# with laundered P / dS bases (parity + timing).
set -u
out=gpurun_out/r42; mkdir -p $out
NAF_FUZZ_CASES=200 timeout 600 python -m pytest tests/test_gpu_fuzz_forward.py -m gpu -q -s -k "5023 or 5058 or 5074 or 5087 or 5110 or 5154 or 5160" > $out/fuzz_outliers.log 2>&1; echo "rc=$?" >> $out/fuzz_outliers.log
grep "^fuzz\|passed\|failed\|^E  " $out/fuzz_outliers.log | cut -c1-400
timeout 900 python -m pytest tests -m gpu -q -k "(test_xna_backward_matches_oracle and (13 or 15)) or test_cell_backward_fuzz or G2-k15" > $out/bwd_chunk_tests.log 2>&1; echo "rc=$?" >> $out/bwd_chunk_tests.log
tail -5 $out/bwd_chunk_tests.log
timeout 300 python tools/bwd_k15_time.py > $out/bwd_k15_time.txt 2>&1; cat $out/bwd_k15_time.txt
