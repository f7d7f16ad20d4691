#!/bin/bash
# probes of merge_u8_val3 at N = 15 (config 4's tile) and N = 7: which pipe bounds it? (measurement builds, wrong results)
set -o pipefail
O=gpurun_out; mkdir -p $O
line() { python -c "import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print(r['kernel'], r['avg_launch_us'], r['frac'], r.get('copy_GBps'))"; }
for w in cfg4tile cfg2 cfg2rand; do
for p in base probe1 probe2 probe3; do
  if [ $p = base ]; then unset HDRMERGE_LIB; else export HDRMERGE_LIB=$PWD/camera_linearity_amd/lib/alt_$p/libhdrmerge.so; fi
  python bench.py --workload $w --no-cpu-baseline --steps 100 > $O/r04b_${w}_$p.log 2>&1; echo -n "$w $p rc=$? "; tail -1 $O/r04b_${w}_$p.log | line
done; done
