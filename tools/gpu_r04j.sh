#!/bin/bash
# round 4, merge_u8_loop rewrite: merge tests, then tools/bench_n.py at N = 17 / 24 / 32 with the shipped library and the alt builds given as arguments
set -o pipefail
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_merge.py -x -q > $O/r04j_pytest_merge.log 2>&1; rc=$?; tail -3 $O/r04j_pytest_merge.log; [ $rc = 0 ] || exit $rc
for rep in 1 2; do
  for alt in main "$@"; do
    if [ $alt = main ]; then unset HDRMERGE_LIB; else export HDRMERGE_LIB=$PWD/camera_linearity_amd/lib/alt_$alt/libhdrmerge.so; fi
    echo "== $alt rep $rep"; timeout -k 10 300 python tools/bench_n.py 16 17 20 24 32 2>&1 | tee -a $O/r04j_bench_n_$alt.log || exit 1
  done
done
