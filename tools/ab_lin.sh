#!/bin/bash
# usage (GPU box): tools/ab_lin.sh <lib name ...>: tools/bench_linearity.py per library
for lib in "$@"; do
  if [ $lib = default ]; then unset HDRMERGE_LIB; else export HDRMERGE_LIB=$PWD/camera_linearity_amd/lib/alt_$lib/libhdrmerge.so; fi
  echo "== $lib"; timeout -k 10 300 python tools/bench_linearity.py 2>&1 | grep use_std
done
