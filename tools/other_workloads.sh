#!/bin/bash
# bench.py on BASELINE's other single-GPU configurations (run on the GPU box); one JSON line per workload
export TMPDIR=/tmp
for w in G2-k7 G2-k11 G2-k15 G3 G4 REF448; do
  python bench.py --workload $w --steps 50 --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1
done
