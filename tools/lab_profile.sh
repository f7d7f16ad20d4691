#!/bin/bash
# PMC passes over the kernel lab (per-variant LDS / wait counters)
export TMPDIR=/tmp
OUT=gpurun_out/${LABOUT:-labprof}; rm -rf $OUT; mkdir -p $OUT
tools/bin/mergelab > $OUT/plain.log 2>&1
for C in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- tools/bin/mergelab > $OUT/pmc_$N.log 2>&1 || echo "pmc $C failed"
done
cat $OUT/plain.log
