#!/bin/bash
# round 4: the new API surface on the GPU - chunked merge (N > 32), statistics over any axis, broadcasting difference / interpolate
set -o pipefail
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_merge.py -x -q -k "chunked or more_than_32 or argument_errors" > $O/r04c_pytest_chunk.log 2>&1; echo "pytest chunk rc=$?"; tail -15 $O/r04c_pytest_chunk.log
timeout -k 10 900 python -m pytest tests/test_gpu_api.py -x -q > $O/r04c_pytest_api.log 2>&1; echo "pytest api rc=$?"; tail -15 $O/r04c_pytest_api.log
