#!/bin/bash
# one workload through tools/profile.sh, summarised on the box (GPU box). usage: tools/gpu_profile_one.sh <tag> <workload> <kernel substring> <algorithmic bytes> <commit>
export TMPDIR=/tmp
O=gpurun_out; TAG=$1; W=$2
mkdir -p $O/${TAG}_profiles
BENCH_ARGS="--workload $W" tools/profile.sh $O/prof_${TAG}_$W > $O/${TAG}_profile_$W.log 2>&1
python3 tools/summarize_profile.py $O/prof_${TAG}_$W ${TAG}_$W "$3" $4 $W $5 > $O/${TAG}_profiles/${TAG}_$W.summary.log 2>&1 || tail -3 $O/${TAG}_profiles/${TAG}_$W.summary.log
rm -rf $O/prof_${TAG}_$W
cp profiles/${TAG}_${W}_rocprof_summary.* profiles/r03_pmc_traffic.json profiles/r03_linearity_valu.json $O/${TAG}_profiles/ 2>/dev/null
grep "algorithmic bytes /" profiles/${TAG}_${W}_rocprof_summary.md
