#!/bin/bash
# round 5, GPU call 6: what do 14 x 14 and 8 x 8 pixel cells cost against 16 x 16 at the reference's timing size (448^2, C = 384, window 9)?
export TMPDIR=/tmp
O=gpurun_out/r05_run6; mkdir -p $O
for rep in 1 2; do for w in REF448 P14 R8; do
  python bench.py --workload $w --steps 300 --no-cpu-baseline --no-live-traffic --no-cold-reading --phase-every 4 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); ph=d['phases_ms']; print('$w', d['ms_per_step'], 'stem', ph.get('stem'), 'rope_pool', ph.get('rope_pool'), 'attention', ph.get('attention'), 'kernel', (d.get('roofline') or {}).get('kernel'))"
done; done 2>&1 | tee $O/cells.txt
