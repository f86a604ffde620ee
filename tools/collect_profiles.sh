#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the default bench + HBM-traffic PMC passes for the
# attention kernel.  Writes under gpurun_out/profiles_raw; summaries are copied into profiles/ by hand.
set -u
export TMPDIR=/tmp
out=gpurun_out/profiles_raw
rm -rf $out; mkdir -p $out
# 1) the bench line itself (un-profiled)
python bench.py > $out/bench_line.json 2> $out/bench.err
tail -1 $out/bench_line.json
# 2) kernel trace + stats of the same command
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py --no-cpu-baseline > $out/trace.log 2>&1
# the JSON line of THAT process: its HIP-event kernel time and the trace average describe the same launches
grep '^{"metric"' $out/trace.log | tail -1 > $out/bench_line_under_rocprof.json
f=$(ls $out/trace/*/*kernel_stats.csv | head -1)
python3 - "$f" > $out/kernel_stats_summary.csv <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
for r in rows[:14]:
    print(",".join(['"%s"' % r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]]))
PY
cat $out/kernel_stats_summary.csv
# 3) HBM traffic of the attention kernel: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slot limit)
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  d=$out/pmc_$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --attention-only > $d.log 2>&1
done
python3 - $out > $out/pmc_summary.txt <<'PY'
import csv, glob, sys, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'xna_mfma' in r['Kernel_Name'] or 'rope_pool' in r['Kernel_Name'] or 'stem_conv' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:80]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()):
        print(f"   {c:28s} launches={len(vals):3d} mean_per_launch={sum(vals)/len(vals):.6g}")
PY
cat $out/pmc_summary.txt
