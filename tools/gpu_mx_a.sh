#!/bin/bash
# MX-fp8 bring-up: layout probe, quantiser, GEMM parity, then the micro-benchmark
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_mxfp8_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/mx_tests.log
cat gpurun_out/mx_tests.log
for tm in 4 5; do
  echo "== MMAE_MX_TM=$tm" >> gpurun_out/mx_bench.log
  MMAE_MX_TM=$tm timeout 300 python tools/mx_gemm_bench.py >> gpurun_out/mx_bench.log 2>&1
done
cat gpurun_out/mx_bench.log
