#!/bin/bash
# usage (GPU box): tools/ab_n.sh "<N list>" <lib ...>  - tools/bench_n.py under each library
ns=$1; shift
for lib in "$@"; do
  if [ $lib = default ]; then unset HDRMERGE_LIB; else export HDRMERGE_LIB=$PWD/camera_linearity_amd/lib/alt_$lib/libhdrmerge.so; fi
  echo "== $lib"; timeout -k 10 300 python tools/bench_n.py $ns 2>&1 | grep "^N="
done
