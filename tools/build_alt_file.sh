#!/bin/bash
# usage: tools/build_alt_file.sh <name> <file.hip> <extra hipcc flags...> -> camera_linearity_amd/lib/alt_<name>/libhdrmerge.so
# (like tools/build_alt.sh, for a source other than hm_merge.hip)
set -e
R=/root/repo; name=$1; src=$2; shift 2
D=$R/camera_linearity_amd/lib/alt_$name; mkdir -p $D
b=$(basename $src .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I$R/include -I$R/camera_linearity_amd/csrc -DHM_TUNE_NF=0 "$@" \
  -c $R/camera_linearity_amd/csrc/$b.hip -o $D/$b.o
objs=$(ls $R/camera_linearity_amd/lib/*.o | grep -v /$b.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs $D/$b.o -o $D/libhdrmerge.so -Wl,-rpath,/opt/rocm/lib
echo built $D/libhdrmerge.so
