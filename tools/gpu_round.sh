#!/bin/bash
# tests + op benches + every bench workload + rocprof passes (GPU box); logs under gpurun_out/<tag>_*
set -o pipefail
TAG=${1:-r2d}; O=gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/${TAG}_pytest_gpu.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"
python tools/bench_ops.py > $O/${TAG}_bench_ops.log 2>&1; cp $O/bench_ops.json $O/${TAG}_bench_ops.json; cut -c1-200 $O/${TAG}_bench_ops.log
python tools/bench_linearity.py 2>&1 | tee $O/${TAG}_bench_linearity.log
python bench.py > $O/${TAG}_bench_default.log 2>&1; tail -1 $O/${TAG}_bench_default.log | cut -c1-1000
for W in ${WORKLOADS:-cfg3 cfg3std cfg3flat cfg3hot cfg4 cfg4std cfg5 cfg2rand cfg2f64 cfg3f64std}; do
  timeout -k 10 600 python bench.py --workload $W --steps 50 --warmup 5 > $O/${TAG}_bench_$W.log 2>&1
  python - <<PY
import json
try:
    d = json.loads(open("$O/${TAG}_bench_$W.log").read().strip().splitlines()[-1])
    c = d.get("cpu_baseline") or {}
    print("$W", d["value"], "Mpix/s", d["roofline"]["avg_launch_us"], "us frac", d["roofline"]["frac"], "|", d["roofline"]["kernel"], "| cpu", c.get("value"), "parity", c.get("parity_ok"), "assembly_ms", d.get("assembly_ms"), flush=True)
except Exception as e:
    print("$W FAILED", e)
PY
done
if [ "${PROFILE:-1}" = "1" ]; then
  cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
  tools/profile.sh $O/prof_${TAG} > $O/${TAG}_profile.log 2>&1; tail -2 $O/${TAG}_profile.log
fi
