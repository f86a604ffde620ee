#!/bin/bash
# round 5, GPU call 8: the de-phased sliding kernel (NAF_XNA_DP=1): parity, then interleaved A/B on the large windows
export TMPDIR=/tmp
O=gpurun_out/r05_run8; mkdir -p $O
NAF_HIP_KNOBS=1 NAF_XNA_DP=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "slid or benched_instantiations or G2_full_size or full_size_properties or fuzz" > $O/pytest_dp.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_dp.txt
tail -6 $O/pytest_dp.txt
for rep in 1 2 3; do for w in G2-k15 G2-k11; do for v in 0 1; do
  NAF_HIP_KNOBS=1 NAF_XNA_DP=$v python bench.py --workload $w --steps 200 --no-cpu-baseline --no-live-traffic --no-cold-reading 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w dp=$v', d['ms_per_step'], r['kernel_ms'], r['frac'], r['kernel'][:60])"
done; done; done 2>&1 | tee $O/ab_dp.txt
