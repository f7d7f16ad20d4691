export TMPDIR=/tmp
for D in 1e-2 5e-2; do
BENCH_ARGS="--workload cfg3hot --hot-density $D --shared-dark" tools/profile.sh gpurun_out/prof_hot > gpurun_out/r03n_profile_hot_$D.log 2>&1
python3 tools/summarize_profile.py gpurun_out/prof_hot r03n_hot${D}_patch merge_patch_hot 1 > /dev/null 2>&1
rm -rf gpurun_out/prof_hot
cp profiles/r03n_hot${D}_patch_rocprof_summary.* gpurun_out/
grep -E "avg_us|FETCH_SIZE|TCC_MISS|TCC_HIT|read " profiles/r03n_hot${D}_patch_rocprof_summary.md | head; python3 -c "
import json; d=json.load(open('profiles/r03n_hot${D}_patch_rocprof_summary.json')); print('$D patch avg us', d['merge_kernel']['avg_us'])"
done
