#!/bin/bash
# round 5, GPU call 1: full GPU suite on the 0.4.0 boundary, stream-layout sweep, timing-only probes of the kernel-free time
export TMPDIR=/tmp
O=gpurun_out/r05_run1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -5 $O/pytest.txt
timeout 600 python tools/streams_crossover.py > $O/streams.txt 2>&1
cat $O/streams.txt
for w in G2-k7 G1; do
  for rep in 1 2; do
    for v in base nomemset; do
      if [ $v = nomemset ]; then e="NAF_HIP_KNOBS=1 NAF_FWD_NO_MEMSET=1"; else e="NAF_X=0"; fi
      env $e python bench.py --workload $w --steps 200 --no-cpu-baseline --no-live-traffic --no-phase-events --no-cold-reading 2>/dev/null | tail -1 | \
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w $v', d['ms_per_step'], d['config'].get('streams'))"
    done
  done
done 2>&1 | tee $O/memset_probe.txt
