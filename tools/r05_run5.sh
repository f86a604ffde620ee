#!/bin/bash
# round 5, GPU call 5: do MFMA and VALU phases of the two waves of a SIMD overlap?  + graph lines for the small workloads
export TMPDIR=/tmp
O=gpurun_out/r05_run5; mkdir -p $O
tools/bin/phase_overlap_probe 2>&1 | tee $O/phase_overlap.txt
for w in G2-k7 S256 G2-k11 G2-k15; do
  for mode in "" "--graph"; do
    python bench.py --workload $w --steps 300 --no-cpu-baseline --no-live-traffic --no-cold-reading $mode 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w $mode', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('kernel_ms'), d['config'].get('streams'))"
  done
done 2>&1 | tee $O/graph_lines.txt
