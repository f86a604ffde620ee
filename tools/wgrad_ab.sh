#!/bin/bash
# Run on the GPU box: the 3x3 / 1x1 weight-gradient kernel (stem_wgrad.hip) -- product against variant libraries given as arguments, interleaved,
# then the LDS counters of the product.   tools/wgrad_ab.sh [variant.so ...]
export TMPDIR=/tmp
out=gpurun_out/wgrad; mkdir -p $out
{
for r in 1 2; do
  python tools/stem_wgrad_bench.py 2>/dev/null | grep -E "TFLOP|plain"
  NAF_HIP_KNOBS=1 NAF_WGRAD_V1=1 python tools/stem_wgrad_bench.py 2>/dev/null | grep -E "TFLOP|plain" | sed 's/^default /round-5 kernel /'
  for v in "$@"; do NAF_HIP_LIB=$PWD/$v python tools/stem_wgrad_bench.py 2>/dev/null | grep -E "TFLOP|plain"; done
done
} | tee $out/ab.txt
if [ "${WGRAD_PMC:-1}" = 1 ]; then
  for lib in product "$@"; do
    [ $lib = product ] && e="NAF_X=0" || e="NAF_HIP_LIB=$PWD/$lib"
    i=0
    for ctrs in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
      i=$((i+1)); d=$out/pmc_$(basename $lib .so)_$i
      (cd /tmp && env $e WGRAD_H=${WGRAD_H:-1024} rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OLDPWD/$d -- python $OLDPWD/tools/stem_wgrad_bench.py > $OLDPWD/$d.log 2>&1) || echo "pass $i failed"
    done
    python3 - $out "$(basename $lib .so)" <<'PY' | tee -a $out/pmc.txt
import csv, glob, sys, collections
out, lib = sys.argv[1:3]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pmc_%s_*/**/*counter_collection.csv' % lib, recursive=True):
    for r in csv.DictReader(open(f)):
        if 'wgrad' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:48] + ' grid ' + r.get('Grid_Size', '?')][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print(lib, k)
    for c, vals in sorted(v.items()):
        print("   %-28s n=%3d mean=%.6g" % (c, len(vals), sum(vals) / len(vals)))
PY
    rm -rf $out/pmc_*_[0-9]
  done
fi
