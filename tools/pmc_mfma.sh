#!/bin/bash
# Matrix-pipe utilisation of every kernel of the forward (run on the GPU box): tools/pmc_mfma.sh <outdir>
#   SQ_VALU_MFMA_BUSY_CYCLES (cycles a SIMD's matrix pipe is busy, summed over the chip), GRBM_GUI_ACTIVE (kernel cycles),
#   SQ_INSTS_MFMA, wave-level wait shares.  GRBM_GUI_ACTIVE comes back summed over the 8 XCDs, so
#   MFMA utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs); effective clock = GUI_ACTIVE / 8 / kernel time.
out=$1
export TMPDIR=/tmp
R=$PWD
mkdir -p $out
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/p$i -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic > $out/p$i.log 2>&1) || echo "pass $i failed"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+'/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if any(k in n for k in ('stem_', 'xna_', 'rope_pool', 'conv0_', 'pack_values')):
            agg[n.split('(')[0][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
print(f"{'kernel':72s} {'launches':>8s} {'cyc/XCD':>10s} {'MFMA busy %':>11s} {'MFMA insts':>11s} {'wait any %':>10s} {'issue stall %':>13s} {'active %':>9s}")
for k,v in agg.items():
    m=lambda c: sum(v[c])/max(1,len(v[c])) if c in v else float('nan')
    cyc=m('GRBM_GUI_ACTIVE')/8.0
    print(f"{k:72s} {len(v['GRBM_GUI_ACTIVE'])//2:8d} {cyc:10.0f} {100*m('SQ_VALU_MFMA_BUSY_CYCLES')/(cyc*256*4):11.1f} {m('SQ_INSTS_MFMA'):11.3g} "
          f"{100*m('SQ_WAIT_ANY')/m('SQ_WAVE_CYCLES'):10.1f} {100*m('SQ_WAIT_INST_ANY')/m('SQ_WAVE_CYCLES'):13.1f} {100*m('SQ_ACTIVE_INST_ANY')/m('SQ_WAVE_CYCLES'):9.1f}")
PY
rm -rf $out/p1 $out/p2
