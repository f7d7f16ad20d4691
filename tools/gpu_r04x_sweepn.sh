#!/bin/bash
# round 4, late: merge_u8_val3 configuration sweep for the frame counts given as arguments (tuning builds alt_tune<N>), 2048 x 4096 x 3, 4 stacks rotating
set -o pipefail
O=gpurun_out; mkdir -p $O
for n in "$@"; do
  export HDRMERGE_LIB=$PWD/camera_linearity_amd/lib/alt_tune$n/libhdrmerge.so
  echo "== N = $n"
  timeout -k 10 400 python tools/ab_val3.py --n $n --h 2048 --w 4096 --stacks 4 --iters 48 --rounds 5 --variants 0,7413,7300,7200,7400,7210,7310,7303,7213,7313,7410,7100 \
    | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print(all(r['bit_equal_to_generic'].values()))
for row in r['rows']: print(row['variant'], row['same_stack_us'], row['rotating_us'], row['rotating_frac'])
"
done 2>&1 | tee $O/r04x_sweepn.log
