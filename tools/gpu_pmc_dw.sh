#!/bin/bash
# PMC pass over the grouped weight-gradient launch: LDS conflicts / waits / MFMA busy
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  rm -rf gpurun_out/pmc_tmp
  (cd /tmp && timeout 300 rocprofv3 --pmc $set -d $R/gpurun_out/pmc_tmp -o p --output-format csv -- python $R/$1 > $R/gpurun_out/pmc_tmp.log 2>&1)
  python tools/pmc_dump.py gpurun_out/pmc_tmp gemm || tail -5 gpurun_out/pmc_tmp.log
done
rm -rf gpurun_out/pmc_tmp
