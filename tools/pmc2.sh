#!/bin/bash
# generic PMC passes: tools/pmc2.sh <outdir> <kernel-substring> -- <cmd...>
out=$1; pat=$2; shift 3
export TMPDIR=/tmp
mkdir -p $out
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_UNALIGNED_STALL" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/p$i -- "$@" > $out/p$i.log 2>&1 || echo "pass $i ($ctrs) failed: $(tail -2 $out/p$i.log)"
done
python3 - "$out" "$pat" <<'PY'
import csv, glob, sys, collections
out,pat=sys.argv[1],sys.argv[2]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+'/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()):
        print(f"   {c:28s} n={len(vals):3d} mean={sum(vals)/len(vals):.6g}")
PY
