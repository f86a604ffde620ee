#!/bin/bash
# Samples the board's power, clocks and temperature (rocm-smi) while a command loops on the GPU.
#   tools/power_sample.sh <label> <command...>
export TMPDIR=/tmp
label=$1; shift
"$@" > /tmp/ps_cmd.log 2>&1 &
pid=$!
sleep 6
echo "== $label"
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp --showperflevel 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (edge|junction|memory)|Performance Level" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'
  echo
  sleep 1.0
done
wait $pid
tail -3 /tmp/ps_cmd.log
