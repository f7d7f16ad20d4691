#!/bin/bash
# GPU-box smoke sequence: full -m gpu suite, smoke(), default bench and the secondary workloads; logs under gpurun_out/<tag>_*
set -o pipefail
TAG=${1:-r2}
O=gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/${TAG}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/${TAG}_smoke.log
python bench.py > $O/${TAG}_bench_default.log 2>&1; echo "bench rc=$?"; tail -1 $O/${TAG}_bench_default.log
for W in ${WORKLOADS:-cfg3 cfg3std cfg3flat cfg3hot cfg4 cfg4std cfg5 cfg2rand}; do
  timeout -k 10 600 python bench.py --workload $W --steps 50 --warmup 5 > $O/${TAG}_bench_$W.log 2>&1; echo "$W rc=$?"; tail -1 $O/${TAG}_bench_$W.log | cut -c1-1500
done
