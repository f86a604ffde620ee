#!/bin/bash
# Run on the GPU box (via gpurun): the evidence files of round 5.
#   1) the default bench line (un-profiled) and the same command under rocprofv3 --kernel-trace --stats
#   2) HBM traffic of the attention kernel for EVERY bench workload: FETCH_SIZE and WRITE_SIZE in separate PMC passes
#      (TCC slot limit), per launch, corrected as MI355X_MICROARCH.md prescribes (gfx950 FETCH_SIZE counts 128-B read
#      requests at 64 B: x2) -> traffic.json
# Writes under gpurun_out/r05prof; the summaries are copied into profiles/ by hand.
set -u
export TMPDIR=/tmp
R=$PWD
out=$R/gpurun_out/r05prof
rm -rf $out; mkdir -p $out
python bench.py > $out/bench_line.json 2> $out/bench.err
python bench.py --no-cpu-baseline --no-live-traffic --phase-every 1 > $out/bench_line_all_phases.json 2>> $out/bench.err
tail -1 $out/bench_line.json | cut -c1-400
# the same step replayed from a hipGraph (launch gaps on record), and the A/B lines of this round's two changes to the forward:
# key pooling on the last stem layers vs the separate pre-pass, two streams vs one (interleaved, one lease)
python bench.py --graph --no-cpu-baseline --no-live-traffic > $out/bench_line_graph.json 2>> $out/bench.err
{
  echo "# interleaved A/B on one lease (bench.py --steps 300 --no-phase-events): stream layout of the one-call forward -- 0 = the library's plan"
  echo "# (naf_forward_streams), 1 = one stream, 2 = the branches side by side on the stream the host lends; ms per step, Mpix/s, streams used"
  for w in G1 G2-k7 S256; do
    for i in 1 2; do
      for v in 0 1 2; do
        python bench.py --workload $w --streams $v --steps 300 --no-phase-events --no-cold-reading --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-6s --streams %d   %.4f ms  %.1f Mpix/s  streams used %s' % ('$w', $v, d['ms_per_step'], d['value'], d['config']['streams']))"
      done
    done
  done
} > $out/ab_streams.txt
cat $out/ab_streams.txt
{
  echo "# bench.py --workload W --steps 100 (one lease): Mpix/s, ms per step, attention kernel ms, fraction of the 8 TB/s HBM roof, of the 2.5 PFLOP/s MFMA roof"
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
cat $out/kernel_stats.csv
for w in G1 G2-k7 G2-k11 G2-k15 G3 G4 REF448; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$out/pmc_${w}_$c
    (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python $R/bench.py --workload $w --steps 3 --warmup 1 --settle-seconds 0 --no-cpu-baseline --no-live-traffic > $d.log 2>&1)
  done
done
python3 - $out > $out/pmc_hbm_traffic.txt <<'PY'
import csv, glob, sys, collections, json, os
out=sys.argv[1]
traffic={}
print("HBM traffic per launch of the attention kernel, per bench workload (rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate passes,")
print("KiB as rocprofv3 reports them; bytes = WRITE_SIZE*1024 + 2*FETCH_SIZE*1024: gfx950 FETCH_SIZE tallies 128-B read requests at 64 B)")
for w in ["G1","G2-k7","G2-k11","G2-k15","G3","G4","REF448"]:
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE","WRITE_SIZE"):
        for f in glob.glob(f"{out}/pmc_{w}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                n=r['Kernel_Name']
                if 'xna_' in n: agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        fs=sum(v['FETCH_SIZE'])/max(1,len(v['FETCH_SIZE'])); ws=sum(v['WRITE_SIZE'])/max(1,len(v['WRITE_SIZE']))
        b=int(ws*1024+2*fs*1024)
        short=k.split('(')[0][:90]
        print(f"{w:8s} {short:92s} launches={len(v['FETCH_SIZE']):2d} FETCH_SIZE={fs:12.1f} WRITE_SIZE={ws:12.1f} -> {b/1e9:.4f} GB")
        traffic[w]={"bytes": b, "kernel": short}
json.dump(traffic, open(os.path.join(out,"traffic.json"),"w"), indent=1)
PY
cat $out/pmc_hbm_traffic.txt
bash tools/pmc_mfma.sh $out/mfma > $out/pmc_mfma_util.txt 2>&1
cat $out/pmc_mfma_util.txt | tail -14
for w in S256 G2-k7; do echo "=== $w"; bash tools/timeline.sh $w; done > $out/timeline_small.txt 2>&1
rm -rf $out/trace $out/pmc_*_FETCH_SIZE $out/pmc_*_WRITE_SIZE $out/mfma
