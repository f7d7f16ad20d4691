#!/bin/bash
# hot-pixel pass against the density of the dark maps (GPU box): cfg3hot with the queue workspace and without it (round 2's pass),
# one map shared by the 7 frames (round 2's setup) and one map per frame. usage: tools/hot_density.sh <out log>
set -o pipefail
OUT=${1:-gpurun_out/hot_density.log}; : > $OUT
for MODE in "--shared-dark" ""; do
  for Q in "" "--no-hot-queue"; do
    for D in 0 1e-4 1e-3 1e-2 5e-2 2e-1; do
      if [ "$Q" = "--no-hot-queue" ] && [ "$D" = "2e-1" ]; then continue; fi
      timeout -k 10 300 python3 bench.py --workload cfg3hot --hot-density $D $MODE $Q --steps 30 --warmup 3 --cpu-rows 64 > gpurun_out/hd.log 2>&1 || { echo "FAILED $D $MODE $Q" >> $OUT; tail -3 gpurun_out/hd.log >> $OUT; continue; }
      python3 - "$D" "$MODE" "$Q" >> $OUT <<'PY'
import json, sys
d = json.loads(open("gpurun_out/hd.log").read().strip().splitlines()[-1])
c = d["cpu_baseline"]
print(f"density {sys.argv[1]:>5} {sys.argv[2] or '--distinct-dark':<16} {sys.argv[3] or '--hot-queue':<15} {d['roofline']['avg_launch_us']:9.1f} us  frac {d['roofline']['frac']:.3f}  "
      f"parity {c['parity_ok']} (val {c['gpu_vs_oracle_max_rel_err']:.1e}, std {c['gpu_vs_oracle_max_rel_err_std']:.1e})  {d['roofline']['kernel']}")
PY
    done
  done
done
cat $OUT
