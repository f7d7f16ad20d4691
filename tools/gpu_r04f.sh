#!/bin/bash
# A/B of std-kernel build variants on the config-3 / config-4 std workloads (one box, one call)
set -o pipefail
O=gpurun_out; mkdir -p $O
line() { python -c "import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print(r['avg_launch_us'], r['frac'], r.get('copy_GBps'))"; }
for rep in 1 2; do
for lib in ${LIBS:-default std_early0 std_fpf0 std_waves2 std_waves4 std_fb1 std_fb4}; do
  if [ $lib = default ]; then unset HDRMERGE_LIB; else export HDRMERGE_LIB=$PWD/camera_linearity_amd/lib/alt_$lib/libhdrmerge.so; fi
  for w in ${WLS:-cfg3flat cfg3std cfg4tilestd}; do
    python bench.py --workload $w --steps 100 --no-cpu-baseline > $O/r04f_${w}_${lib}_$rep.log 2>&1; echo -n "$rep $lib $w rc=$? "; tail -1 $O/r04f_${w}_${lib}_$rep.log | line
  done
done; done
