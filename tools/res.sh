#!/bin/bash
# usage: tools/res.sh <file.hip> [pattern]  - VGPR / SGPR / scratch / occupancy of every kernel in one source file
R=/root/repo
f=$1; pat=${2:-k_}
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I$R/include -I$R/camera_linearity_amd/csrc -DHM_TUNE_NF=${TUNE_NF:-0} \
  -Rpass-analysis=kernel-resource-usage -c $R/camera_linearity_amd/csrc/$f -o /tmp/res_$$.o 2> /tmp/res_$$.txt
grep -i "error" /tmp/res_$$.txt | head
python $R/tools/resusage.py /tmp/res_$$.txt "$pat"
rm -f /tmp/res_$$.o /tmp/res_$$.txt
