#!/bin/bash
# usage (GPU box): tools/ab_var.sh <rounds> "<variants>" <lib ...>   - cfg2 launch time per (library, variant)
rounds=$1; vars=$2; shift 2
for r in $(seq $rounds); do
  for lib in "$@"; do
    if [ $lib = default ]; then unset HDRMERGE_LIB; else export HDRMERGE_LIB=$PWD/camera_linearity_amd/lib/alt_$lib/libhdrmerge.so; fi
    for v in $vars; do
      timeout -k 10 200 python bench.py --no-cpu-baseline --steps 200 --warmup 20 --variant $v > gpurun_out/ab.log 2>&1
      python - <<PY
import json;d=json.loads(open("gpurun_out/ab.log").read().strip().splitlines()[-1]);print("$lib","variant $v",d["roofline"]["avg_launch_us"],d["roofline"]["frac"],flush=True)
PY
    done
  done
done
