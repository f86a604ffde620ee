#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) of the product library and a variant: tools/ab_kernels.sh VARIANT.so WORKLOAD
V=$1; W=${2:-G1}
cd /tmp && export TMPDIR=/tmp
for lib in product variant; do
  if [ $lib = variant ]; then export NAF_HIP_LIB=$GRAFT_REPO_ROOT/$V; else unset NAF_HIP_LIB; fi
  rm -rf /tmp/abk_$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk_$lib -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 50 --no-cpu-baseline --no-live-traffic > /tmp/abk_$lib.log 2>&1
  echo "== $lib $(grep '^{' /tmp/abk_$lib.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"; tail -3 /tmp/abk_$lib.log | cut -c1-200
  f=$(find /tmp/abk_$lib -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("  %-90s calls %5s avg %9.1f us  total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
