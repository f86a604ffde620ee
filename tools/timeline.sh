#!/bin/bash
# Timeline of one forward step from rocprofv3 --kernel-trace: start offset, duration, stream (queue) of every dispatch and the time no
# kernel was running.  tools/timeline.sh WORKLOAD [bench args]
W=${1:-G1}; shift
export TMPDIR=/tmp
R=$PWD
rm -rf /tmp/tl; (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --no-phase-events "$@" > /tmp/tl.log 2>&1)
grep '^{"metric"' /tmp/tl.log | tail -1 | cut -c1-160
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps are delimited by the attention kernel (last kernel of a forward)
ends = [i for i, r in enumerate(rows) if "xna_" in r["Kernel_Name"]]
if len(ends) < 12: print("too few steps", len(ends)); sys.exit()
for which in (len(ends) // 2, len(ends) // 2 + 1):
    a, b = ends[which - 1] + 1, ends[which]
    step = rows[a:b + 1]
    t0 = int(rows[ends[which - 1]]["End_Timestamp"])       # end of the previous step's last kernel
    print("--- step %d: %d dispatches, previous attention end -> this attention end %.1f us" % (which, len(step), (int(step[-1]["End_Timestamp"]) - t0) / 1e3))
    busy_end = t0; idle = 0.0
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - busy_end) / 1e3
        if gap > 0: idle += gap
        print("  +%8.1f us  %7.1f us  q%-3s gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), gap, r["Kernel_Name"].split("(")[0][:70]))
        busy_end = max(busy_end, e)
    print("  no kernel running: %.1f us" % idle)
PY
