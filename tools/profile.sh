#!/bin/bash
# rocprofv3 passes over the bench (run on the GPU box): kernel trace + stats, then PMC passes (own runs,
# no tracing domains beside --kernel-trace). Outputs under gpurun_out/prof_*; summaries are copied to profiles/ by
# tools/summarize_profile.py afterwards.
set -o pipefail
export TMPDIR=/tmp
OUT=${1:-gpurun_out/prof}
PROG=${PROG:-bench.py}                     # another program (tools/prof_weight.py ...) takes no bench arguments
ARGS="--steps 30 --warmup 5 --no-cpu-baseline ${BENCH_ARGS}"
if [ "$PROG" != "bench.py" ]; then ARGS="${BENCH_ARGS}"; fi
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $PROG $ARGS > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES" "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $PROG $ARGS > $OUT/pmc_$N.log 2>&1 || { echo "pmc $C failed"; tail -3 $OUT/pmc_$N.log; }
done
find $OUT -name "*.csv" | head -40
