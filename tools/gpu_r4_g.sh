#!/bin/bash
# round 4, visit g: fp16-storage ('h16') adapter -- kernel tests, adapter parity, bench A/B against the default ('f16') mode
mkdir -p gpurun_out
export TMPDIR=/tmp
S=gpurun_out/r4g_summary.txt
: > $S
timeout 900 python -m pytest tests/test_h16_gpu.py -q --tb=short -p no:cacheprovider > gpurun_out/r4g_pytest.log 2>&1
tail -40 gpurun_out/r4g_pytest.log | grep -E "passed|failed|Error|assert|FAILED" | head -30 >> $S
timeout 600 python -m pytest tests/test_parity_geometry_gpu.py -q --tb=short -p no:cacheprovider -k "fp16_storage or bench_geometry" > gpurun_out/r4g_pytest_p.log 2>&1
tail -25 gpurun_out/r4g_pytest_p.log | grep -E "passed|failed|Error|assert|FAILED" | head -20 >> $S
cp gpurun_out/grad_parity_cfg3_bf16_h16.txt gpurun_out/r4g_grad_parity_cfg3_bf16_h16.txt 2>/dev/null
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5"
run() { label=$1; shift; ( "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-60)" >> $S; tail -3 gpurun_out/x.err | grep -i "error\|Traceback" >> $S; }
run "f16 (default)" timeout 300 $B
run "h16" timeout 300 $B --fp32-adapter-gemm h16
run "f16 (again)" timeout 300 $B
run "h16 (again)" timeout 300 $B --fp32-adapter-gemm h16
cat $S
