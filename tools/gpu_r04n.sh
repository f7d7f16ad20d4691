#!/bin/bash
# round 4: statistics with infinite terms (NumPy's answers) - the new test, every statistics / linearity test, the fuzz, and the linearity benches
set -o pipefail
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_producers.py tests/test_reference_suite.py -x -q -m gpu > $O/r04n_pytest_api.log 2>&1; rc=$?; tail -4 $O/r04n_pytest_api.log; [ $rc = 0 ] || exit $rc
timeout -k 10 300 python tools/fuzz_backends.py --seconds 100 --seed 31 --log $O/r04n_fuzz_seed31.log > /dev/null 2>&1; rc=$?; tail -4 $O/r04n_fuzz_seed31.log; [ $rc = 0 ] || exit $rc
line() { python -c "import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print(r['avg_launch_us'], r['frac'], (l.get('roofline_valu') or {}).get('frac'), (l.get('cpu_baseline') or {}).get('parity_ok'), r.get('traffic_stale'))"; }
for rep in 1 2; do for w in linearity linearitystd; do
  python bench.py --workload $w --steps 30 > $O/r04n_bench_${w}_$rep.log 2>&1; echo -n "$w rep $rep rc=$? "; tail -1 $O/r04n_bench_${w}_$rep.log | line
done; done
timeout -k 10 300 python tools/bench_r04_new.py 2>&1 | grep -i "axis" | tee $O/r04n_bench_axis.log
