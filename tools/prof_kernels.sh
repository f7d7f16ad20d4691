#!/bin/bash
# PMC passes over tools/prof_kernels.py (GPU box): VALU / LDS / wait cycles of k_welford and k_energy_partial.
set -o pipefail
export TMPDIR=/tmp
OUT=${1:-gpurun_out/prof_kernels}
rm -rf $OUT; mkdir -p $OUT
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  echo "pass $N"
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 tools/prof_kernels.py > $OUT/pmc_$N.log 2>&1 || { echo "pmc $C failed"; tail -3 $OUT/pmc_$N.log; }
done
find $OUT -name "*counter_collection.csv" | head
