#!/bin/bash
# usage (GPU box): tools/ab_var2.sh <workload> "<variants>" <lib>   - launch time per variant under one library
w=$1; vars=$2; lib=$3
if [ $lib != default ]; then export HDRMERGE_LIB=$PWD/camera_linearity_amd/lib/alt_$lib/libhdrmerge.so; fi
for v in $vars; do
  timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --steps 100 --warmup 10 --variant $v > gpurun_out/ab.log 2>&1
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab.log").read().strip().splitlines()[-1]);print("$w variant $v",d["roofline"]["avg_launch_us"],d["roofline"]["frac"],flush=True)
except Exception as e:
    print("$w variant $v failed", open("gpurun_out/ab.log").read()[-300:])
PY
done
