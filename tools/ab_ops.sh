#!/bin/bash
# usage (GPU box): tools/ab_ops.sh <grep pattern> <lib name ...>  ("default" = the in-tree library): tools/bench_ops.py per library, matching rows
pat=$1; shift
for lib in "$@"; do
  if [ $lib = default ]; then unset HDRMERGE_LIB; else export HDRMERGE_LIB=$PWD/camera_linearity_amd/lib/alt_$lib/libhdrmerge.so; fi
  timeout -k 10 300 python tools/bench_ops.py > gpurun_out/ab_ops_$lib.log 2>&1
  echo "== $lib"; grep -i "$pat" gpurun_out/ab_ops_$lib.log | cut -c1-200
done
