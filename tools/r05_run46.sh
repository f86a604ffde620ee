#!/bin/bash
# Round 5, end: training-path fuzz incl. the differentiable HIP stem; rocprofv3 kernel stats of the backward bench.
set -u
export TMPDIR=/tmp
R=$PWD
out=$R/gpurun_out/r46; mkdir -p $out
NAF_FUZZ_TRAIN_CASES=80 timeout 900 python -m pytest tests/test_gpu_fuzz_train.py -m gpu -q -s > $out/fuzz_train.log 2>&1; echo "rc=$?" >> $out/fuzz_train.log
grep -c "train fuzz" $out/fuzz_train.log; grep "^E  .*train fuzz\|passed\|failed" $out/fuzz_train.log | cut -c1-500 | head -20
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $R/tools/xna_bwd_bench.py > $out/bwd_bench.log 2>&1)
f=$(ls $out/trace/*/*kernel_stats.csv | head -1)
python3 - "$f" > $out/bwd_kernel_stats.csv <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
for r in rows[:14]:
    print(",".join(['"%s"' % r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]]))
PY
cat $out/bwd_kernel_stats.csv | cut -c1-220; tail -12 $out/bwd_bench.log
rm -rf $out/trace
