#!/bin/bash
# usage (GPU box): tools/ab3.sh <rounds> <workload> <lib name ...>   ("default" = the in-tree library) - interleaved rounds
rounds=$1; w=$2; shift 2
for r in $(seq $rounds); do
  for lib in "$@"; do
    if [ $lib = default ]; then unset HDRMERGE_LIB; else export HDRMERGE_LIB=$PWD/camera_linearity_amd/lib/alt_$lib/libhdrmerge.so; fi
    timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/ab.log 2>&1
    python - <<PY
import json;d=json.loads(open("gpurun_out/ab.log").read().strip().splitlines()[-1]);print("$w","$lib",d["roofline"]["avg_launch_us"],d["roofline"]["frac"],flush=True)
PY
  done
done
