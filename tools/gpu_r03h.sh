timeout -k 10 800 python -m pytest tests -q -m gpu -x > gpurun_out/r03h_pytest_gpu.log 2>&1; echo pytest rc=$?; tail -3 gpurun_out/r03h_pytest_gpu.log | cut -c1-300
python3 tools/bench_ops.py > gpurun_out/r03h_bench_ops.log 2>&1; cp gpurun_out/bench_ops.json gpurun_out/r03h_bench_ops.json; grep -i "hot_pixel\|pair_stat\|channel_stat" gpurun_out/r03h_bench_ops.log | cut -c1-200
for W in linearity linearitystd; do timeout -k 10 600 python3 bench.py --workload $W > gpurun_out/r03h_bench_$W.log 2>&1; tail -1 gpurun_out/r03h_bench_$W.log | cut -c1-1500; done
HDRMERGE_LIB=camera_linearity_amd/lib/alt_tune7/libhdrmerge.so python3 tools/ab_val3.py --variants 0,67413,97413,127413,157413,187413,247413 --rounds 5 --out gpurun_out/r03h_ab_val3_wg.json | tail -1 | cut -c1-1600
