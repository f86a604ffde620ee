#!/bin/bash
# round 5, GPU call 10: G3 (C = 1024, 1024^2) by images per launch: what one rank of the 8-GPU job runs
export TMPDIR=/tmp
O=gpurun_out/r05_run10; mkdir -p $O
for b in 1 2 4 8; do
  python bench.py --workload G3 --per-gpu-batch $b --steps 40 --no-cpu-baseline --no-live-traffic --no-cold-reading 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('G3 B=$b', d['value'], 'Mpix/s', d['ms_per_step'], 'ms/step', (d.get('roofline') or {}).get('kernel_ms'), d['config'].get('streams'))"
done 2>&1 | tee $O/g3_batch.txt
