#!/bin/bash
# usage (GPU box): tools/sweep_env.sh <ENV_NAME> "<values>" <rounds> <workload ...>  - interleaved rounds of bench.py with ENV_NAME=value (0 = unset)
E=$1; VALS=$2; R=$3; shift 3
for r in $(seq $R); do
  for w in "$@"; do
    for v in $VALS; do
      if [ "$v" = "0" ]; then unset $E; else export $E=$v; fi
      timeout -k 10 200 python3 bench.py --workload $w --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/sw.log 2>&1 || { echo "$w $E=$v FAILED"; continue; }
      python3 -c "
import json
d=json.loads(open('gpurun_out/sw.log').read().strip().splitlines()[-1])
print('$w $E=$v', d['roofline']['avg_launch_us'], 'us frac', d['roofline']['frac'])"
    done
  done
done
