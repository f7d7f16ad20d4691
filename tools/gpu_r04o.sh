#!/bin/bash
# A/B on one box: builds of hm_stats.hip (alt_<name>) against the working tree on linearitystd (and linearity), repetitions interleaved
set -o pipefail
O=gpurun_out; mkdir -p $O
line() { python -c "import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print(r['avg_launch_us'], (l.get('cpu_baseline') or {}).get('parity_ok'))"; }
for rep in 1 2; do for alt in "$@" new; do
  if [ $alt = new ]; then unset HDRMERGE_LIB; else export HDRMERGE_LIB=$PWD/camera_linearity_amd/lib/alt_$alt/libhdrmerge.so; fi
  for w in ${W:-linearitystd}; do
    python bench.py --workload $w --steps 30 > $O/r04o_${w}_${alt}_$rep.log 2>&1; echo -n "$alt $w rep $rep rc=$? "; tail -1 $O/r04o_${w}_${alt}_$rep.log | line
  done
done; done 2>&1 | tee -a $O/r04o_ab_stats_inf.log
