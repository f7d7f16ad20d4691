#!/bin/bash
# rocprofv3 over tools/bench_producers.py (GPU box): kernel trace + stats, then two PMC passes for the HBM traffic of
# k_welford. Output under gpurun_out/prof_producers; tools/summarize_producers.py copies the summary to profiles/.
set -o pipefail
export TMPDIR=/tmp
OUT=${1:-gpurun_out/prof_producers}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_producers.py --quick > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
for C in ; do   # PMC passes disabled: the 0.5 s pre-warm loops issue ~1e5 tiny launches, each serialised by counter collection
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 tools/bench_producers.py --quick > $OUT/pmc_$C.log 2>&1 || { echo "pmc $C failed"; tail -3 $OUT/pmc_$C.log; }
done
find $OUT -name "*.csv" | head
