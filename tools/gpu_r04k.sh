#!/bin/bash
# round 4, merge_u8_loop rewrite, second pass: occupancy targets (alt builds) and workgroups per CU (HM_TUNE_LOOP_WG_PER_CU in the env build)
set -o pipefail
O=gpurun_out; mkdir -p $O
run() { echo "== $1 wg=${HM_TUNE_LOOP_WG_PER_CU:-auto}"; timeout -k 10 300 python tools/bench_n.py 17 24 32 $2 2>&1 | grep "^N=" | tee -a $O/r04k_bench_n.log; }
for rep in 1 2; do
  unset HDRMERGE_LIB HM_TUNE_LOOP_WG_PER_CU; run main --val-only || exit 1
  for alt in old w8 w6 w5; do export HDRMERGE_LIB=$PWD/camera_linearity_amd/lib/alt_$alt/libhdrmerge.so; run $alt --val-only || exit 1; done
  export HDRMERGE_LIB=$PWD/camera_linearity_amd/lib/alt_env/libhdrmerge.so
  for wg in 7 14 21 28; do export HM_TUNE_LOOP_WG_PER_CU=$wg; run env --val-only || exit 1; done
  unset HM_TUNE_LOOP_WG_PER_CU
done
