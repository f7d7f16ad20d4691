#!/bin/bash
# final verification of a round (GPU box): tests, smoke, standalone kernels, the default bench line, the N > 1 rehearsal (two ranks sharing
# the one GPU, gloo carrying the barrier, no torchrun on the command line). usage: tools/gpu_check.sh <tag>
set -o pipefail
TAG=${1:-r03k}; O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -1 $O/${TAG}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/${TAG}_smoke.log
python tools/bench_ops.py > $O/${TAG}_bench_ops.log 2>&1; cp $O/bench_ops.json $O/${TAG}_bench_ops.json; echo "bench_ops rc=$?"
python bench.py > $O/${TAG}_bench_default.log 2>&1; echo "bench rc=$?"; tail -1 $O/${TAG}_bench_default.log | cut -c1-700
python bench.py --gpus 2 --share-device --dist-backend gloo > $O/${TAG}_bench_2ranks.log 2>&1; echo "2 ranks rc=$?"; tail -1 $O/${TAG}_bench_2ranks.log | cut -c1-400
python bench.py --gpus 2 --share-device --dist-backend gloo --workload cfg4 > $O/${TAG}_bench_cfg4_2ranks.log 2>&1; echo "cfg4 2 ranks rc=$?"; tail -1 $O/${TAG}_bench_cfg4_2ranks.log | cut -c1-400
