#!/bin/bash
# round 4: the unit-size / prefetch / workgroups-per-CU sweep of merge_u8_val3 for N = 15 on config 4's tile (1024 x 8192 x 3), tuning build
# (TUNE_NF=15 tools/build_alt.sh tune15), same process, shared output buffers, 4 stacks rotating. Variants 7UPM; W*10000 + 7UPM sets W workgroups per CU.
set -o pipefail
O=gpurun_out; mkdir -p $O
export HDRMERGE_LIB=$PWD/camera_linearity_amd/lib/alt_tune15/libhdrmerge.so
timeout -k 10 500 python tools/ab_val3.py --n 15 --h 1024 --w 8192 --stacks 4 --iters 64 --rounds 5 \
  --variants 0,7300,7100,7200,7400,7600,7800,7310,7210,7410,7303,7313,7213,47300,57300,67300,107300,127300,167300,87200,127200,87400,127400 --out $O/r04x_sweep15.json \
  | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print(r['bit_equal_to_generic'])
for row in r['rows']: print(row['variant'], row['same_stack_us'], row['rotating_us'], row['rotating_frac'])
" | tee $O/r04x_sweep15.log
