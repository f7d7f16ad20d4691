#!/bin/bash
# round 4, first GPU call: the never-run paths (nccl world of one, shared-memory assembly, device guard), the negative-input parity
# test, and same-box baselines of the kernels this round works on.
set -o pipefail
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_multiprocess.py -x -q > $O/r04a_pytest_mp.log 2>&1; echo "pytest mp rc=$?"; tail -3 $O/r04a_pytest_mp.log
timeout -k 10 600 python -m pytest tests/test_gpu_merge.py -x -q -k "negative or float or linearize" > $O/r04a_pytest_neg.log 2>&1; echo "pytest neg rc=$?"; tail -3 $O/r04a_pytest_neg.log
python bench.py --gpus 2 --share-device --dist-backend gloo --workload cfg4 --no-cpu-baseline > $O/r04a_cfg4_2ranks.log 2>&1; echo "cfg4 2 ranks rc=$?"; tail -1 $O/r04a_cfg4_2ranks.log | cut -c1-1500
python bench.py --gpus 2 --share-device --dist-backend gloo --workload cfg4std --no-cpu-baseline > $O/r04a_cfg4std_2ranks.log 2>&1; echo "cfg4std 2 ranks rc=$?"; tail -1 $O/r04a_cfg4std_2ranks.log | cut -c1-1500
for w in cfg4tile cfg4tilestd cfg3std cfg3flat linearitystd; do
  python bench.py --workload $w --no-cpu-baseline > $O/r04a_bench_$w.log 2>&1; echo "$w rc=$?"; tail -1 $O/r04a_bench_$w.log | python -c "import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print(r['kernel'], r['avg_launch_us'], r['frac'], (l.get('roofline_valu') or {}).get('frac'))"
done
