#!/bin/bash
# tests + op benches + A/B of the float64-frame std kernel + default bench + rocprof passes (GPU box)
set -o pipefail
TAG=${1:-r2c}; O=gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/${TAG}_pytest_gpu.log | cut -c1-300
python tools/bench_ops.py > $O/${TAG}_bench_ops.log 2>&1; cp $O/bench_ops.json $O/${TAG}_bench_ops.json; cut -c1-200 $O/${TAG}_bench_ops.log
python tools/bench_linearity.py 2>&1 | tee $O/${TAG}_bench_linearity.log
tools/ab3.sh 2 cfg3f64std default keepw3 keepw4 f64w3 2>&1 | tee $O/${TAG}_ab_f64std.log
python bench.py > $O/${TAG}_bench_default.log 2>&1; tail -1 $O/${TAG}_bench_default.log | cut -c1-1200
python bench.py --workload cfg3 --steps 50 --warmup 5 > $O/${TAG}_bench_cfg3.log 2>&1; tail -1 $O/${TAG}_bench_cfg3.log | cut -c1-1200
python bench.py --workload cfg4 --steps 50 --warmup 5 > $O/${TAG}_bench_cfg4.log 2>&1; tail -1 $O/${TAG}_bench_cfg4.log | cut -c1-1200
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
tools/profile.sh $O/prof_${TAG} > $O/${TAG}_profile.log 2>&1; tail -3 $O/${TAG}_profile.log
