#!/bin/bash
# Memory-path PMC passes (run on the GPU box): tools/pmc_mem.sh <outdir> <cmd...>
# L2 -> fabric request counts / stalls, L1 -> L2 request latencies, TLB stalls, wave-level wait breakdown, per kernel name.
out=$1; shift
export TMPDIR=/tmp
mkdir -p $out
i=0
for ctrs in \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
  "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
  "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_NORMAL_WRITEBACK_sum" \
  "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
  "TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_PENDING_STALL_CYCLES_sum" \
  "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/p$i -- "$@" > $out/p$i.log 2>&1) || echo "pass $i ($ctrs) failed: $(tail -2 $out/p$i.log)"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
dur=collections.defaultdict(list)
for f in glob.glob(out+'/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob(out+'/p*/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Kernel_Name']].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
for k,v in agg.items():
    d=dur.get(k,[0])
    print(f"{k[:150]}   [{len(d)} launches, mean {sum(d)/len(d):.4f} ms under PMC]")
    for c,vals in sorted(v.items()):
        print(f"   {c:40s} n={len(vals):3d} mean={sum(vals)/len(vals):.6g}")
PY
