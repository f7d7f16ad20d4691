#!/bin/bash
# verification run of a round (GPU box): tests, smoke, standalone kernels, the default bench line and the secondary workloads, the N > 1
# rehearsals (two ranks sharing the one GPU over gloo; one nccl rank). usage: tools/gpu_check.sh <tag>
set -o pipefail
TAG=${1:-r04z}; O=gpurun_out; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -1 $O/${TAG}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/${TAG}_smoke.log
python tools/bench_ops.py > $O/${TAG}_bench_ops.log 2>&1; rc=$?; cp $O/bench_ops.json $O/${TAG}_bench_ops.json; echo "bench_ops rc=$rc"
python bench.py > $O/${TAG}_bench_default.log 2>&1; echo "bench rc=$?"; tail -1 $O/${TAG}_bench_default.log | cut -c1-900
for W in ${BENCHES:-cfg3 cfg3std cfg3flat cfg3hot cfg4 cfg4std cfg5 cfg2rand cfg2f64 cfg3f64std linearity linearitystd welford energy}; do
  timeout -k 10 400 python bench.py --workload $W --steps 100 --warmup 10 > $O/${TAG}_bench_$W.log 2>&1; echo -n "$W rc=$? "
  tail -1 $O/${TAG}_bench_$W.log | python -c "import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print(r['avg_launch_us'], r['frac'], (l.get('roofline_valu') or {}).get('frac'), (l.get('cpu_baseline') or {}).get('parity_ok'), l.get('assembly_ms'))"
done
python bench.py --gpus 2 --share-device --dist-backend gloo > $O/${TAG}_bench_2ranks.log 2>&1; echo "2 ranks rc=$?"; tail -1 $O/${TAG}_bench_2ranks.log | cut -c1-400
python bench.py --gpus 2 --share-device --dist-backend gloo --workload cfg4 > $O/${TAG}_bench_cfg4_2ranks.log 2>&1; echo "cfg4 2 ranks rc=$?"; tail -1 $O/${TAG}_bench_cfg4_2ranks.log | cut -c1-400
python bench.py --gpus 1 --force-dist --dist-backend nccl --workload cfg4 --no-cpu-baseline > $O/${TAG}_bench_cfg4_nccl1.log 2>&1; echo "cfg4 nccl world 1 rc=$?"; tail -1 $O/${TAG}_bench_cfg4_nccl1.log | cut -c1-400
