#!/bin/bash
# samples rocm-smi while a command runs: scripts/power_probe.sh <out> <cmd...>   (is the step power-limited? clocks and power under load)
out=$1; shift
"$@" > /dev/null 2>&1 &
pid=$!
sleep 20
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showmaxpower --showuse --showtemp 2>/dev/null | grep -v "^=\|^$" >> $out
  echo "----" >> $out
  sleep 2
done
wait $pid
