#!/bin/bash
# Run on the GPU box (via gpurun): the evidence files of round 6 on the FINAL tree.  Writes under gpurun_out/r06; the summaries are copied
# into profiles/ by hand (profiles/README.md).
#   1 the default bench line (un-profiled) + the same command under rocprofv3 --kernel-trace --stats
#   2 the other workloads' table, with the sliding kernel's tail hand-over A/B at G2-k11 / a 13 x 13 window (NAF_XNA_STEAL=1)
#   3 the attention backward: event-timed lines of every window + rocprofv3 --kernel-trace --stats of the SHIPPED dispatch
#   4 the reference's backward protocol in fp32 and under autocast (the HIP stem is the training default)
set -u
export TMPDIR=/tmp
R=$PWD
out=$R/gpurun_out/r06
rm -rf $out; mkdir -p $out
stats_csv() {   # $1 = rocprofv3 output directory, $2 = rows
python3 - "$(ls $1/*/*kernel_stats.csv | head -1)" "$2" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
for r in rows[:int(sys.argv[2])]:
    print(",".join(['"%s"' % r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]]))
PY
}
# ---- 1
python bench.py > $out/bench_line.json 2> $out/bench.err
tail -1 $out/bench_line.json | cut -c1-300
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $R/bench.py --no-cpu-baseline --no-live-traffic > $out/trace.log 2>&1)
grep '^{"metric"' $out/trace.log | tail -1 > $out/bench_line_under_rocprof.json
stats_csv $out/trace 16 > $out/kernel_stats.csv; rm -rf $out/trace
cut -c1-160 $out/kernel_stats.csv | head -8
# ---- 2
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-22s %8.2f Mpix/s  %.4f ms/step  attention %.4f ms  hbm %.4f  mfma %.4f  stem %.4f' % (sys.argv[1], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r['mfma_frac'], d['phases_ms']['stem']))" "$1"; }
{
  echo "# bench.py --workload W --steps 100 (one lease, final tree of round 6): Mpix/s, ms per step, attention kernel ms, fraction of the 8 TB/s HBM roof, of the 2.5 PFLOP/s MFMA roof"
  for w in G2-k7 G2-k11 G2-k15 G3 G4 REF448 S256; do
    python bench.py --workload $w --steps 100 --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 | line $w
  done
  echo "# the sliding-window kernel's tail hand-over (xna_slide_kernel.h STEAL; VERDICT r05 item 2) against the static split, interleaved, 400 steps each"
  for i in 1 2 3; do
    for st in 0 1; do
      NAF_HIP_KNOBS=1 NAF_XNA_STEAL=$st python bench.py --workload G2-k11 --steps 400 --no-cpu-baseline --no-live-traffic --no-cold-reading 2>/dev/null | tail -1 | line "G2-k11 steal=$st"
    done
  done
} > $out/other_workloads.txt
cat $out/other_workloads.txt
# ---- 3
{
  echo "# tools/bwd_k15_time.py --fast (event-timed, three repetitions): the attention backward per window on the final tree"
  for i in 1 2 3; do python tools/bwd_k15_time.py --fast 2>/dev/null; done
} > $out/bwd.txt
cat $out/bwd.txt | tail -9
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_bwd -- python $R/tools/bwd_k15_time.py --fast > $out/trace_bwd.log 2>&1)
stats_csv $out/trace_bwd 14 > $out/bwd_kernel_stats.csv; rm -rf $out/trace_bwd
cut -c1-170 $out/bwd_kernel_stats.csv
# ---- 4
python tools/backward_speed_protocol.py > $out/backward_speed_protocol.txt 2>&1
cat $out/backward_speed_protocol.txt
