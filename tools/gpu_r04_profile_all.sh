#!/bin/bash
# round-4 counter profiles (GPU box): tools/profile.sh per workload, summarised ON the box into profiles/<TAG>_<workload>_rocprof_summary.{md,json}
# and profiles/r04_pmc_counters.json (copied back under gpurun_out/<TAG>_profiles/). usage: tools/gpu_r04_profile_all.sh <commit> [workloads...]
export TMPDIR=/tmp
O=gpurun_out; H=${1:-?}; shift; TAG=${TAG:-r04p}
mkdir -p $O/${TAG}_profiles
declare -A K=( [cfg2]=merge_u8_val3 [cfg2rand]=merge_u8_val3 [cfg3]=merge_u8_fast_std+merge_scan_hot+merge_patch_hot [cfg3std]=merge_u8_fast_std [cfg3flat]=merge_u8_fast_std
               [cfg4tile]=merge_u8_val3 [cfg4tilestd]=merge_u8_fast_std [cfg2f64]=merge_f64_val [cfg3f64std]=merge_f64_std [linearity]=k_pairs_stats [linearitystd]=k_pairs_stats
               [welford]=k_welford [energy]=k_energy [cfg3hot]=merge_u8_fast_std+merge_scan_hot+merge_patch_hot [cfg5]=merge_u8_val3 )
declare -A B=( [cfg2]=754974720 [cfg2rand]=754974720 [cfg3]=4781506560 [cfg3std]=3976200192 [cfg3flat]=4429185024 [cfg4tile]=578813952 [cfg4tilestd]=3800039424
               [cfg2f64]=3221225472 [cfg3f64std]=6442450944 [linearity]=2818572288 [linearitystd]=5637144576 [welford]=3221225472 [energy]=550502400 [cfg3hot]=4328521728 [cfg5]=754974720 )
for W in ${@:-cfg2 cfg3 cfg3std cfg3flat cfg4tile cfg4tilestd linearity linearitystd welford energy}; do
  BENCH_ARGS="--workload $W" tools/profile.sh $O/prof_${TAG}_$W > $O/${TAG}_profile_$W.log 2>&1
  python3 tools/summarize_profile.py $O/prof_${TAG}_$W ${TAG}_$W "${K[$W]}" ${B[$W]} $W $H > $O/${TAG}_profiles/${TAG}_$W.summary.log 2>&1 || { echo "summary $W FAILED"; tail -3 $O/${TAG}_profiles/${TAG}_$W.summary.log; }
  rm -rf $O/prof_${TAG}_$W
  grep "algorithmic bytes /" profiles/${TAG}_${W}_rocprof_summary.md
done
cp profiles/${TAG}_* profiles/r04_pmc_counters.json $O/${TAG}_profiles/ 2>/dev/null
du -sh $O
