#!/bin/bash
# usage: tools/build_alt.sh <name> <extra hipcc flags...>  -> camera_linearity_amd/lib/alt_<name>/libhdrmerge.so
# A second build of the library that differs only in hm_merge.hip's compile flags (A/B runs on one box via HDRMERGE_LIB).
set -e
R=/root/repo; name=$1; shift
D=$R/camera_linearity_amd/lib/alt_$name; mkdir -p $D
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I$R/include -I$R/camera_linearity_amd/csrc -DHM_TUNE_NF=${TUNE_NF:-0} "$@" \
  -c $R/camera_linearity_amd/csrc/hm_merge.hip -o $D/hm_merge.o
objs=$(ls $R/camera_linearity_amd/lib/*.o | grep -v hm_merge.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs $D/hm_merge.o -o $D/libhdrmerge.so -Wl,-rpath,/opt/rocm/lib
echo built $D/libhdrmerge.so
