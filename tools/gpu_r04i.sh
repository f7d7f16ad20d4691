#!/bin/bash
# A/B: the dark-map scan beside the streaming kernel (side stream) against behind it, one box, one call
set -o pipefail
O=gpurun_out; mkdir -p $O
export HDRMERGE_LIB=$PWD/camera_linearity_amd/lib/alt_sideenv/libhdrmerge.so
line() { python -c "import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print(r['avg_launch_us'], r['frac'], (l.get('cpu_baseline') or {}).get('parity_ok'))"; }
for rep in 1 2; do
for side in 0 1; do
  if [ $side = 1 ]; then export HM_TUNE_SCAN_SIDE=1; else unset HM_TUNE_SCAN_SIDE; fi
  for w in cfg3 cfg3hot; do
    python bench.py --workload $w --steps 100 --cpu-rows 128 > $O/r04i_${w}_side${side}_$rep.log 2>&1; echo -n "$rep side=$side $w rc=$? "; tail -1 $O/r04i_${w}_side${side}_$rep.log | line
  done
  python bench.py --workload cfg3hot --hot-density 1e-2 --steps 50 --cpu-rows 128 > $O/r04i_cfg3hot1e2_side${side}_$rep.log 2>&1; echo -n "$rep side=$side cfg3hot@1e-2 rc=$? "; tail -1 $O/r04i_cfg3hot1e2_side${side}_$rep.log | line
done; done
