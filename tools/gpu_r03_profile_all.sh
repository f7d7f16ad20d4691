#!/bin/bash
# every round-3 counter profile in one go (GPU box): tools/profile.sh per workload, summarised ON the box (the raw per-dispatch CSVs of
# eleven workloads exceed what gpurun copies back), summaries collected under gpurun_out/r03i_profiles/; then a plain bench run of each workload.
# usage: tools/gpu_r03_profile_all.sh <commit>
export TMPDIR=/tmp
O=gpurun_out; H=${1:-?}; TAG=${TAG:-r03i}
mkdir -p $O/${TAG}_profiles
prof() {   # workload kernel-substring algorithmic-bytes
  BENCH_ARGS="--workload $1" tools/profile.sh $O/prof_${TAG}_$1 > $O/${TAG}_profile_$1.log 2>&1
  python3 tools/summarize_profile.py $O/prof_${TAG}_$1 ${TAG}_$1 "$2" $3 $1 $H > $O/${TAG}_profiles/${TAG}_$1.summary.log 2>&1 || { echo "summary $1 FAILED"; tail -3 $O/${TAG}_profiles/${TAG}_$1.summary.log; }
  rm -rf $O/prof_${TAG}_$1
  grep "algorithmic bytes /" profiles/${TAG}_$1_rocprof_summary.md
}
prof cfg3 merge_u8_fast_std+merge_scan_hot+merge_patch_hot 4781506560
if [ -z "$ONLY" ]; then
prof cfg2 merge_u8_val3 754974720
prof cfg3std merge_u8_fast_std 3976200192
prof cfg3flat merge_u8_fast_std 4429185024
prof cfg4tile merge_u8_val3 578813952
prof cfg2rand merge_u8_val3 754974720
prof cfg3f64std merge_f64_std 6442450944
prof cfg2f64 merge_f64_val 3221225472
prof linearity k_pairs_stats_lds 2818572288
prof linearitystd k_pairs_stats_lds 5637144576
PROG=tools/prof_weight.py BENCH_ARGS="" tools/profile.sh $O/prof_${TAG}_weightf64 > $O/${TAG}_profile_weightf64.log 2>&1
python3 tools/summarize_profile.py $O/prof_${TAG}_weightf64 ${TAG}_weightf64 k_weight_f64 1207959552 > $O/${TAG}_profiles/${TAG}_weightf64.summary.log 2>&1; rm -rf $O/prof_${TAG}_weightf64
grep "algorithmic bytes /" profiles/${TAG}_weightf64_rocprof_summary.md
fi
cp profiles/${TAG}_* profiles/r03_pmc_traffic.json profiles/r03_linearity_valu.json $O/${TAG}_profiles/ 2>/dev/null
for W in ${BENCHES:-cfg2 cfg3std cfg3flat cfg3 cfg3hot cfg4tile cfg4 cfg4std cfg5 cfg2rand cfg3f64std cfg2f64 linearity linearitystd}; do
  timeout -k 10 400 python3 bench.py --workload $W --steps 100 --warmup 10 > $O/${TAG}_bench_$W.log 2>&1; echo "$W rc=$?"; tail -1 $O/${TAG}_bench_$W.log | cut -c1-120
done
du -sh $O
