#!/bin/bash
# pre-split x3 products of the fp32 output adapter: parity suite of the model, then A/B of the step
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_parity_geometry_gpu.py -x -q 2>&1 | grep -v "^    \|^$" | tail -8
: > gpurun_out/x3.log
run() { label="$1"; shift; echo "== $label" >> gpurun_out/x3.log; env "$@" > gpurun_out/x.log 2> gpurun_out/x.err; grep "timed region" gpurun_out/x.err | tail -1 | cut -c1-120 >> gpurun_out/x3.log; tail -1 gpurun_out/x.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print({k:r.get(k) for k in ('achieved','gemm_ms_per_step','f32_adapter_gemm_ms_per_step','f32_adapter_gemm_tflops')})" >> gpurun_out/x3.log 2>&1; }
B="timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5"
run "cfg3 pre-split x3 (default)" A=1 $B
run "cfg3 MMAE_X3_PRESPLIT=0" MMAE_X3_PRESPLIT=0 $B
run "cfg3 pre-split x3, again" A=1 $B
run "cfg3 MMAE_X3_PRESPLIT=0, again" MMAE_X3_PRESPLIT=0 $B
cat gpurun_out/x3.log
