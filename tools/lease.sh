#!/bin/bash
# One parametrised lease script (replaces the per-call tools/r05_run*.sh transcripts): tools/lease.sh <step> [<step> ...]
# Every step writes under gpurun_out/<step>/ and prints a short tail; steps are independent.
#   gpurun --timeout 1500 -- 'bash tools/lease.sh wino tests'
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/.."; R=$PWD
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
    train)       # round 6: the training path on the HIP stem (every width), the device-evaluated oracle cases, the 8-rank rehearsal
      timeout 1400 python -m pytest tests/test_gpu_train_stem.py tests/test_gpu_fuzz_train.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q -s --durations=25 \
        -k "train or backward or oracle_on_the_device or F8 or denoising or G4 or G3 or dry_run or differentiable or plain_convolution or act_ or weight_gradient or first_convolution or strided" > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
      grep -E "denoising training step|passed|failed|rc=|Error|error" $out/tests.log | tail -30; tail -45 $out/tests.log | cut -c1-220 ;;
    steal)       # round 6 item 2: the sliding-window kernel's tail hand-over -- parity first, then interleaved A/B against the static split
      timeout 900 python -m pytest tests -m gpu -q -x -k "G2_full_size or sliding_window or benched_instantiations or full_size_properties or fuzz_forward" > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log; tail -4 $out/tests.log | cut -c1-200
      for w in G2-k11 G2-k13x; do
        [ $w = G2-k13x ] && continue
        for i in 1 2 3; do
          for st in 1 0; do
            NAF_HIP_KNOBS=1 NAF_XNA_STEAL=$st python bench.py --workload $w --steps 400 --no-cpu-baseline --no-live-traffic --no-cold-reading 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); r = j['roofline']
print('$w steal=$st  step %.4f ms  attention %.4f ms  frac %.4f  mfma_frac %.4f' % (j['ms_per_step'], r['kernel_ms'], r['frac'], r['mfma_frac']))"
          done
        done
      done | tee $out/ab.txt ;;
    collect)     # the round's evidence files (tools/collect_profiles_r06.sh -> gpurun_out/r06)
      bash tools/collect_profiles_r06.sh 2>&1 | tail -80 | cut -c1-220 ;;
    fuzz)        # campaigns of the two fuzzers (more cases than the suite's defaults): the training path with the model width drawn per case
      NAF_FUZZ_TRAIN_CASES=${NAF_FUZZ_TRAIN_CASES:-120} NAF_FUZZ_TRAIN_SEED=${NAF_FUZZ_TRAIN_SEED:-7200} timeout 1200 python -m pytest tests/test_gpu_fuzz_train.py -m gpu -q -s > $out/train.log 2>&1; echo "rc=$?" >> $out/train.log
      grep -cE "^.?train fuzz" $out/train.log; tail -3 $out/train.log | cut -c1-200
      NAF_FUZZ_CASES=${NAF_FUZZ_CASES:-150} NAF_FUZZ_SEED=${NAF_FUZZ_SEED:-5600} timeout 1200 python -m pytest tests/test_gpu_fuzz_forward.py -m gpu -q -s > $out/forward.log 2>&1; echo "rc=$?" >> $out/forward.log
      tail -3 $out/forward.log | cut -c1-200 ;;
    denoise)     # the denoising training step at its model widths: HIP stem (auto) vs the torch arms, with a kernel table
      python tools/denoise_train_time.py --profile > $out/denoise_train.txt 2>&1; grep -E "^NAF|stem_|xna_|Self CUDA" $out/denoise_train.txt | cut -c1-260 ;;
    gen)         # the general-width stem kernels after a change: their parity tests, then the denoising training step
      timeout 900 python -m pytest tests -m gpu -q -x -k "any_width or plain_convolution or weight_gradient or act_backward or first_convolution or without_autocast or denoising or F2 or F6 or F7 or fuzz_train" > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log; tail -4 $out/tests.log | cut -c1-200
      python tools/denoise_train_time.py --profile > $out/denoise_train.txt 2>&1; grep -E "^NAF|stem_|Self CUDA time" $out/denoise_train.txt | cut -c1-100,200-260 ;;
    driver)      # what the driver does at round end: smoke(), the GPU suite, the default bench line
      python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
      timeout 1400 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log; tail -4 $out/tests.log | cut -c1-200
      python bench.py --steps 20 > $out/bench20.json 2> $out/bench20.err; python3 -c "
import json; j = json.load(open('$out/bench20.json')); r = j['roofline']
print('bench --steps 20: %.2f Mpix/s  %.4f ms (from idle %.4f)  attention %.4f ms = %.4f of the roof; first steps settled %s | from idle %s' % (j['value'], j['ms_per_step'], j['ms_per_step_no_settle'], r['kernel_ms'], r['frac'], j['step_ms']['head'][:4], j['step_ms_no_settle']['head'][:6]))" ;;
    bwdpt)       # round 6: partial row tiles in the cell backward (patch-14 cells) -- parity, then the patch-14 training point's backward
      timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz_train.py tests/test_gpu_fullsize.py -m gpu -q -x -s -k "backward or fuzz_train or denoising" > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log; grep -aE "passed|failed|rc=|Error" $out/tests.log | tail -5 | cut -c1-200
      python tools/bwd_k15_time.py 2>/dev/null | grep -E "448|k  9" | tee $out/p14.txt ;;
    bwdfuzz)     # the cell-backward fuzz as a campaign (partial row tiles included)
      NAF_FUZZ_BWD_SEED=${NAF_FUZZ_BWD_SEED:-6161} NAF_FUZZ_BWD_CASES=${NAF_FUZZ_BWD_CASES:-300} timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "test_cell_backward_fuzz" > $out/bwd.log 2>&1; echo "rc=$?" >> $out/bwd.log
      grep -ac "^bwd fuzz" $out/bwd.log; grep -aE "passed|failed|rc=" $out/bwd.log | tail -3 ;;
    trainprof)   # which kernels a training step of the reference's backward protocol spends its time in (torch.profiler table, three steps)
      python tools/backward_speed_protocol.py --profile > $out/profile.txt 2>&1; grep -aE "^REF448|^P14" $out/profile.txt | cut -c1-200
      (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace -- python $R/tools/backward_speed_protocol.py > $R/$out/trace.log 2>&1)
      python3 - "$(ls $out/trace/*/*kernel_stats.csv | head -1)" <<'PY' | tee $out/kernel_stats.csv | cut -c1-180 | head -50
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("Name,Calls,TotalDurationNs,AverageNs,Percentage")
for r in rows[:60]:
    print(",".join(['"%s"' % r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]]))
PY
      rm -rf $out/trace ;;
    *) echo "unknown step $step" ;;
  esac
done
