#!/bin/bash
# Run on the GPU box: one bench line per BASELINE workload (whole forward + its attention kernel), one lease.
echo "python bench.py --workload <W> --no-cpu-baseline --steps 20, one lease; whole forward and its attention kernel"
for w in G1 G2-k7 G2-k11 G2-k15 G3 G4 REF448; do
  python bench.py --workload $w --no-cpu-baseline --steps 20 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-8s %7.1f Mpix/s  %7.3f ms/step  attention %.4f ms  %5.0f GB/s  frac %.3f  traffic/algorithmic %.3f  MFMA %.0f TFLOP/s  %s' % (
    '$w', d['value'], d['ms_per_step'], r['kernel_ms'], r['achieved'], r['frac'], (r['traffic'] or 0) / r['algorithmic_bytes'], r.get('mfma_tflops', 0), r['kernel'][:48]))"
done
python bench.py --workload G1 --per-gpu-batch 16 --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('G1, 16 images per step: %.1f Mpix/s, %.2f ms/step' % (d['value'], d['ms_per_step']))"
