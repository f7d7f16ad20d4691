python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
python bench.py > gpurun_out/bench_cfg2.log 2>&1; tail -1 gpurun_out/bench_cfg2.log
python bench.py --workload cfg3 > gpurun_out/bench_cfg3.log 2>&1; tail -1 gpurun_out/bench_cfg3.log
python bench.py --workload cfg4tile > gpurun_out/bench_cfg4tile.log 2>&1; tail -1 gpurun_out/bench_cfg4tile.log
