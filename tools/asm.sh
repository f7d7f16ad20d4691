#!/bin/bash
# tools/asm.sh <file.hip> [extra flags]: device-only assembly of one source -> /tmp/<name>.s, prints VGPR counts of its kernels
R=/root/repo
f=$1; shift
n=$(basename $f .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I$R/include -I$R/camera_linearity_amd/csrc -DHM_TUNE_NF=${TUNE_NF:-0} "$@" \
  -Wall -Wno-unused-function -S --cuda-device-only $R/camera_linearity_amd/csrc/$n.hip -o /tmp/$n.s 2>&1 | grep -v "warning: argument unused"
grep -E "^\s*\.set .*\.(num_vgpr|private_seg_size)," /tmp/$n.s | sed 's/^\s*\.set //' 
