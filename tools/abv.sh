#!/bin/bash
# usage (GPU box): tools/abv.sh <rounds> <lib name|default> <workload> <variant ...>  - interleaved rounds of bench.py --variant V
rounds=$1; lib=$2; w=$3; shift 3
if [ $lib = default ]; then unset HDRMERGE_LIB; else export HDRMERGE_LIB=$PWD/camera_linearity_amd/lib/alt_$lib/libhdrmerge.so; fi
for r in $(seq $rounds); do
  for v in "$@"; do
    timeout -k 10 200 python bench.py --workload $w --variant $v --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/abv.log 2>&1
    python - <<PY
import json;d=json.loads(open("gpurun_out/abv.log").read().strip().splitlines()[-1]);print("$w","$lib","variant $v",d["roofline"]["avg_launch_us"],d["roofline"]["frac"],d["roofline"].get("kernel"),flush=True)
PY
  done
done
