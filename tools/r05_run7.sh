#!/bin/bash
# round 5, GPU call 7: the full GPU suite and smoke() on the final code state
export TMPDIR=/tmp
O=gpurun_out/r05_run7; mkdir -p $O
timeout 2000 python -m pytest tests -m gpu -q --durations=12 > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -20 $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-700 | tee $O/bench20.txt
