#!/bin/bash
# usage (GPU box): tools/clocks_during_bench.sh <lib name|default> <workload> <variant> <steps>
# polls rocm-smi (sclk, socket power; fclk / mclk are fixed at 1250 / 2000 MHz on these boxes) while bench.py runs <steps> launches
lib=$1; w=$2; v=$3; steps=$4
if [ $lib = default ]; then unset HDRMERGE_LIB; else export HDRMERGE_LIB=$PWD/camera_linearity_amd/lib/alt_$lib/libhdrmerge.so; fi
python bench.py --workload $w --variant $v --no-cpu-baseline --steps $steps --warmup 20 > gpurun_out/b_long.log 2>&1 &
pid=$!
sleep 4
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|Power (W)" | sed 's/.*: //' | tr -s '\t ' ' ' | paste -sd' '
  sleep 1
done
wait $pid
python - <<PY
import json;d=json.loads(open("gpurun_out/b_long.log").read().strip().splitlines()[-1]);print("$lib $w variant $v:",d["roofline"]["avg_launch_us"],"us",d["roofline"]["frac"])
PY
