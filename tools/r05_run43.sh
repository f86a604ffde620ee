#!/bin/bash
# Round 5, late: fuzz campaigns -- the seven outliers again (operands rounded), a second whole-forward seed range, the training path.
set -u
out=gpurun_out/r43; mkdir -p $out
NAF_FUZZ_CASES=200 timeout 300 python -m pytest tests/test_gpu_fuzz_forward.py -m gpu -q -s -k "5023 or 5058 or 5074 or 5087 or 5110 or 5154 or 5160" > $out/fuzz_outliers.log 2>&1; echo "rc=$?" >> $out/fuzz_outliers.log
grep -o "fuzz 5.*" $out/fuzz_outliers.log | cut -c1-330; tail -2 $out/fuzz_outliers.log
NAF_FUZZ_TRAIN_CASES=120 timeout 1200 python -m pytest tests/test_gpu_fuzz_train.py -m gpu -q -s > $out/fuzz_train.log 2>&1; echo "rc=$?" >> $out/fuzz_train.log
tail -12 $out/fuzz_train.log | cut -c1-400
NAF_FUZZ_SEED=6000 NAF_FUZZ_CASES=300 timeout 900 python -m pytest tests/test_gpu_fuzz_forward.py -m gpu -q -s > $out/fuzz_forward2.log 2>&1; echo "rc=$?" >> $out/fuzz_forward2.log
tail -12 $out/fuzz_forward2.log | cut -c1-400
