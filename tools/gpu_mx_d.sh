#!/bin/bash
# MX-fp8 evidence: per-product micro-benchmark, MFMA-pipe PMC of the same launches, kernel stats of a cfg5 step in both modes
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/mx_gemm_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/mx_gemm_bench.txt
rm -rf gpurun_out/pmc_mx
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_mx -o p --output-format csv -- python $R/tools/mx_gemm_bench.py > $R/gpurun_out/pmc_mx.log 2>&1)
python tools/pmc_mfma.py gpurun_out/pmc_mx > gpurun_out/pmc_mfma_mx.txt 2>&1
rm -rf gpurun_out/pmc_mx
for prec in mxfp8 bf16; do
  rm -rf gpurun_out/prof_cfg5_$prec
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cfg5_$prec -o p --output-format csv -- python $R/bench.py --config cfg5 --precision $prec --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_cfg5_$prec.log 2>&1)
  f=$(find gpurun_out/prof_cfg5_$prec -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_cfg5_$prec.csv
  rm -rf gpurun_out/prof_cfg5_$prec
done
cat gpurun_out/mx_gemm_bench.txt gpurun_out/pmc_mfma_mx.txt
head -12 gpurun_out/kernel_stats_cfg5_mxfp8.csv | cut -c1-150
