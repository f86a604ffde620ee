#!/bin/bash
# rocprofv3 kernel trace of the table-driven attention kernels on their use cases (run on the GPU box via gpurun).
set -u
export TMPDIR=/tmp
out=gpurun_out/profiles_table
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/sweep -- python tools/shape_sweep.py > $out/sweep.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/denoise -- python tools/denoise_shape_time.py > $out/denoise.log 2>&1
for d in sweep denoise; do
  f=$(ls $out/$d/*/*kernel_stats.csv | head -1)
  echo "== $d"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("Name,Calls,AverageNs,MinNs,MaxNs")
for r in rows:
    if any(k in r["Name"] for k in ("xna_union", "xna_rows", "xna_generic", "xna_mfma", "xna_slide", "axis_table")):
        print(",".join(['"%s"' % r["Name"][:100], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"]]))
PY
done | tee $out/summary.csv
