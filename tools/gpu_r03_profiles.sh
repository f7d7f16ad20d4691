#!/bin/bash
# round-3 counter evidence (GPU box): tools/profile.sh over the workloads named on the command line, then a plain bench run of each
# on the same box. usage: tools/gpu_r03_profiles.sh <tag> <workload ...>     (logs under gpurun_out/<tag>_*, profiles under gpurun_out/prof_<tag>_<workload>)
set -o pipefail
TAG=$1; shift
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for W in "$@"; do
  BENCH_ARGS="--workload $W" tools/profile.sh $O/prof_${TAG}_$W > $O/${TAG}_profile_$W.log 2>&1; echo "profile $W rc=$?"
  timeout -k 10 300 python3 bench.py --workload $W --steps 100 --warmup 10 --no-cpu-baseline > $O/${TAG}_bench_$W.log 2>&1
  tail -1 $O/${TAG}_bench_$W.log | cut -c1-400
done
