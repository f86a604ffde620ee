#!/bin/bash
# Half-row staging of the cell kernel (xna_mfma_kernel HS, round 6): parity of the shapes that take it, then interleaved A/B against whole-row staging
# (NAF_XNA_HS=0) on the workloads it changes: G2-k7 and G3 (Dv = 256 at 7 x 7), G1-k9 / G3-k9 (the reference's default window).
export TMPDIR=/tmp
out=gpurun_out/hs; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -x -k "xna_mfma_matches_oracle or benched_instantiations or full_size_properties or G2_full_size or G3 or fuzz_forward or return_weights or rotate_on_load or F5 or F10" 2>&1 | grep -aE "passed|failed|Error" | tail -3
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-22s %8.2f Mpix/s  %.4f ms/step  attention %.4f ms  hbm %.4f' % (sys.argv[1], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac']))" "$1"; }
for rep in 1 2 3; do
  for w in G2-k7 G3 G1-k9 G3-k9 G1; do
    for hs in 1 0; do
      NAF_HIP_KNOBS=1 NAF_XNA_HS=$hs python bench.py --workload $w --steps 300 --no-cpu-baseline --no-live-traffic --no-cold-reading 2>/dev/null | tail -1 | line "$w hs=$hs"
    done
  done
done | tee $out/ab.txt
