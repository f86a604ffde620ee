#!/bin/bash
# Interleaved A/B of the product library against a variant on one lease: tools/ab_lib.sh VARIANT.so "G1 G2-k7 ..." [rounds] [steps]
V=$1; WL=${2:-G1}; R=${3:-3}; S=${4:-100}
for w in $WL; do for r in $(seq $R); do for lib in product variant; do
  if [ $lib = variant ]; then e="NAF_HIP_LIB=$V"; else e="NAF_X=0"; fi
  env $e python bench.py --workload $w --steps $S --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w $lib', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done; done
