#!/bin/bash
# Run on the GPU box (via gpurun): the evidence files of round 5 on the FINAL tree -- the default bench line (un-profiled), the same command under
# rocprofv3 --kernel-trace --stats, the other workloads' table.  (The forward's kernels did not change after tools/collect_profiles_r05.sh ran; the PMC
# passes and A/B sweeps of that script are not repeated.)  Writes under gpurun_out/r05final; the summaries are copied into profiles/ by hand.
set -u
export TMPDIR=/tmp
R=$PWD
out=$R/gpurun_out/r05final
rm -rf $out; mkdir -p $out
python bench.py > $out/bench_line.json 2> $out/bench.err
tail -1 $out/bench_line.json | cut -c1-400
{
  echo "# bench.py --workload W --steps 100 (one lease, final tree of round 5): Mpix/s, ms per step, attention kernel ms, fraction of the 8 TB/s HBM roof, of the 2.5 PFLOP/s MFMA roof"
  for w in G2-k7 G2-k11 G2-k15 G3 G4 REF448 S256; do
    python bench.py --workload $w --steps 100 --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-8s %8.2f Mpix/s  %.4f ms/step  attention %.4f ms  hbm %.4f  mfma %.4f  stem %.4f' % ('$w', d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r['mfma_frac'], d['phases_ms']['stem']))"
  done
} > $out/other_workloads.txt
cat $out/other_workloads.txt
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $R/bench.py --no-cpu-baseline --no-live-traffic > $out/trace.log 2>&1)
grep '^{"metric"' $out/trace.log | tail -1 > $out/bench_line_under_rocprof.json
f=$(ls $out/trace/*/*kernel_stats.csv | head -1)
python3 - "$f" > $out/kernel_stats.csv <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
for r in rows[:16]:
    print(",".join(['"%s"' % r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]]))
PY
cat $out/kernel_stats.csv | cut -c1-200
rm -rf $out/trace
