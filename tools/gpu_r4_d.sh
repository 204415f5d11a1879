#!/bin/bash
# Round-4 visit D: full GPU suite on the current build + smoke + bench lines.
mkdir -p gpurun_out
export TMPDIR=/tmp
S=gpurun_out/r4d_summary.txt
: > $S
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r4d_pytest_gpu.log 2>&1
tail -6 gpurun_out/r4d_pytest_gpu.log >> $S
grep -E "^FAILED|^ERROR" gpurun_out/r4d_pytest_gpu.log | head -20 >> $S
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-220 >> $S
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5"
run() { label=$1; shift; ( "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-120)" >> $S; tail -2 gpurun_out/x.err | grep -i "error\|Traceback" >> $S; }
run "cfg3 default (f16 adapter products + f16 attention cores)" timeout 300 $B
run "cfg3 x3" timeout 300 $B --fp32-adapter-gemm x3
run "cfg3 default (again)" timeout 300 $B
head -8 gpurun_out/grad_parity_cfg3_bf16.txt >> $S
grep -n "semseg" gpurun_out/grad_parity_cfg3_bf16.txt | head -5 >> $S
cat $S
