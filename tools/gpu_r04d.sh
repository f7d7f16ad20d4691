#!/bin/bash
# A/B of the all-pairs statistics kernel variants (one box, one call): bench.py linearity / linearitystd per library
set -o pipefail
O=gpurun_out; mkdir -p $O
line() { python -c "import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print(r['avg_launch_us'], (l.get('cpu_baseline') or {}).get('parity_ok'), (l.get('cpu_baseline') or {}).get('gpu_vs_oracle_max_rel_err'))"; }
for rep in 1 2; do
for lib in ${LIBS:-lin_base default lin_mb16 lin_mb32 lin_lean lin_lean16}; do
  if [ $lib = default ]; then unset HDRMERGE_LIB; else export HDRMERGE_LIB=$PWD/camera_linearity_amd/lib/alt_$lib/libhdrmerge.so; fi
  for w in linearitystd linearity; do
    python bench.py --workload $w --steps 60 --cpu-rows 64 > $O/r04d_${w}_${lib}_$rep.log 2>&1; echo -n "$rep $lib $w rc=$? "; tail -1 $O/r04d_${w}_${lib}_$rep.log | line
  done
done; done
