#!/bin/bash
# usage (GPU box): tools/power_poll.sh <command...> - runs the command in the background and polls rocm-smi (sclk, socket power) once a second
"$@" > gpurun_out/power_cmd.log 2>&1 &
pid=$!
sleep 2
for i in 1 2 3 4 5; do
  rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|Power (W)" | sed 's/.*: //' | tr -s '\t ' ' ' | paste -sd' '
  sleep 1
done
wait $pid; cat gpurun_out/power_cmd.log
