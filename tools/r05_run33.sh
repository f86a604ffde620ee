#!/bin/bash
# round 5, GPU call 33: LDS-DMA staging of the cell kernel's windows and RoPE tables (-DNAF_XNA_GLDS=1 variant) -- parity, then interleaved A/B of the forward
export TMPDIR=/tmp
O=gpurun_out/r05_run33; mkdir -p $O
NAF_HIP_LIB=$PWD/tools/bin/libnaf_glds.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "mfma or golden or benched or G1 or fuzz or logits or rotate" 2>&1 | tail -4 | tee $O/pytest.txt
for r in 1 2 3; do
  for w in G1 G3 G4 S256; do
    echo "== $w product"; timeout 300 python bench.py --workload $w --steps 200 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
    echo "== $w LDS-DMA";  NAF_HIP_LIB=$PWD/tools/bin/libnaf_glds.so timeout 300 python bench.py --workload $w --steps 200 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done 2>&1 | tee $O/ab.txt
