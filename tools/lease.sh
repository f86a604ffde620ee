#!/bin/bash
# One parametrised lease script (replaces the per-call tools/r05_run*.sh transcripts): tools/lease.sh <step> [<step> ...]
# Every step writes under gpurun_out/<step>/ and prints a short tail; steps are independent.
#   gpurun --timeout 1500 -- 'bash tools/lease.sh wino tests'
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for step in "$@"; do
  out=gpurun_out/$step; mkdir -p "$out"
  case $step in
    wino)        # the Winograd F(2x2,3x3) instruction mix, with the board's power and clocks beside it
      tools/bin/winograd_probe 64 > $out/probe.txt 2>&1; cat $out/probe.txt
      for m in 1 5 3; do bash tools/power_sample.sh "winograd probe mode $m" tools/bin/winograd_probe 4000 $m; done > $out/power.txt 2>&1; cat $out/power.txt ;;
    tests)       # the whole GPU suite with per-test durations
      timeout 1400 python -m pytest tests -m gpu -x -q --durations=70 > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log; tail -90 $out/tests.log | cut -c1-220 ;;
    bench)       # the driver's line
      python bench.py > $out/bench.json 2> $out/bench.err; tail -c 3000 $out/bench.json ;;
    *) echo "unknown step $step" ;;
  esac
done
