#!/bin/bash
# round 5, GPU call 2: full GPU suite; memset-free start A/B (valid data on both arms)
export TMPDIR=/tmp
O=gpurun_out/r05_run2; mkdir -p $O
timeout 2000 python -m pytest tests -m gpu -q --durations=25 > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -40 $O/pytest.txt
for w in G2-k7 S256 G1; do
  for rep in 1 2 3; do
    for v in base memset; do
      if [ $v = memset ]; then e="NAF_HIP_KNOBS=1 NAF_FWD_MEMSET=1"; else e="NAF_X=0"; fi
      env $e python bench.py --workload $w --steps 300 --no-cpu-baseline --no-live-traffic --no-phase-events --no-cold-reading 2>/dev/null | tail -1 | \
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w $v', d['ms_per_step'], d['config'].get('streams'))"
    done
  done
done 2>&1 | tee $O/memset_ab.txt
