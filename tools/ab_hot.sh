#!/bin/bash
# usage (GPU box): tools/ab_hot.sh <rounds> "<bench args>" <lib ...>   - cfg3hot with the given arguments, interleaved over alternate builds
rounds=$1; args=$2; shift 2
for r in $(seq $rounds); do
  for lib in "$@"; do
    if [ $lib = default ]; then unset HDRMERGE_LIB; else export HDRMERGE_LIB=$PWD/camera_linearity_amd/lib/alt_$lib/libhdrmerge.so; fi
    timeout -k 10 300 python3 bench.py --workload cfg3hot $args --no-cpu-baseline --steps 40 --warmup 4 > gpurun_out/abh.log 2>&1
    python3 - <<PY
import json;d=json.loads(open("gpurun_out/abh.log").read().strip().splitlines()[-1]);print("cfg3hot $args","$lib",d["roofline"]["avg_launch_us"],flush=True)
PY
  done
done
