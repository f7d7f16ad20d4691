#!/bin/bash
# usage (on the GPU box): tools/ab.sh <alt name> <rounds> <workload...>   - interleaved A/B of the default and an alternate library
alt=$1; rounds=$2; shift 2
for r in $(seq $rounds); do
  for w in "$@"; do
    for lib in default $alt; do
      if [ $lib = default ]; then unset HDRMERGE_LIB; else export HDRMERGE_LIB=$PWD/camera_linearity_amd/lib/alt_$lib/libhdrmerge.so; fi
      timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/ab.log 2>&1
      python - <<PY
import json;d=json.loads(open("gpurun_out/ab.log").read().strip().splitlines()[-1]);print("$w","$lib",d["ms_per_step"],d["roofline"]["frac"],flush=True)
PY
    done
  done
done
