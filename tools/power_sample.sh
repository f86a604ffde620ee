#!/bin/bash
# Samples the board's power, core clock and temperatures (rocm-smi) while a command loops on the GPU.
#   tools/power_sample.sh <label> <command...>
export TMPDIR=/tmp
label=$1; shift
"$@" > /tmp/ps_cmd.log 2>&1 &
pid=$!
sleep 6
echo "== $label"
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | awk '
    /Socket Graphics Package Power/ {p=$NF}
    /sclk clock level/ {gsub(/[()]/,"",$NF); s=$NF}
    /Sensor junction/ {j=$NF}
    /Sensor memory/ {m=$NF}
    END {printf "   power %s W   sclk %s   junction %s C   memory %s C\n", p, s, j, m}'
  sleep 1.0
done
wait $pid
tail -1 /tmp/ps_cmd.log | cut -c1-160
