#!/bin/bash
set -o pipefail
O=gpurun_out; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > $O/r04e_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/r04e_pytest_gpu.log
python tools/bench_ops.py > $O/r04e_bench_ops.log 2>&1; rc=$?; cp $O/bench_ops.json $O/r04e_bench_ops.json; echo "bench_ops rc=$rc"
for w in welford energy linearitystd linearity; do python bench.py --workload $w > $O/r04e_bench_$w.log 2>&1; echo "$w rc=$?"; tail -1 $O/r04e_bench_$w.log | cut -c1-1800; done
