#!/bin/bash
# MX-fp8 encoder mode: parity tests, then cfg5 / cfg3 bf16 vs mxfp8 on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mxfp8_gpu.py -x -q 2>&1 | grep -v "^    \|^$" | tail -25 > gpurun_out/mx_tests.log
cat gpurun_out/mx_tests.log
: > gpurun_out/mx_step.log
for cfg in cfg5 cfg3; do
  for prec in bf16 mxfp8; do
    echo "== $cfg $prec" >> gpurun_out/mx_step.log
    timeout 600 python bench.py --config $cfg --precision $prec --no-cpu-baseline --steps 15 --warmup 4 > gpurun_out/mx_${cfg}_${prec}.log 2> gpurun_out/mx_${cfg}_${prec}.err
    grep "timed region" gpurun_out/mx_${cfg}_${prec}.err | tail -1 | cut -c1-200 >> gpurun_out/mx_step.log
    tail -1 gpurun_out/mx_${cfg}_${prec}.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print({k:r.get(k) for k in ('achieved','frac','gemm_ms_per_step','launches_per_step')}, r.get('mxfp8'))" >> gpurun_out/mx_step.log 2>&1
  done
done
cat gpurun_out/mx_step.log
