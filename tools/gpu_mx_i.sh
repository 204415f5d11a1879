#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mxfp8_gpu.py -x -q 2>&1 | grep -v "^    \|^$" | tail -6
: > gpurun_out/mx_step3.log
run() { label="$1"; shift; echo "== $label" >> gpurun_out/mx_step3.log; env "$@" > gpurun_out/x.log 2> gpurun_out/x.err; grep "timed region" gpurun_out/x.err | tail -1 | cut -c1-200 >> gpurun_out/mx_step3.log; }
B="timeout 600 python bench.py --no-cpu-baseline --steps 15 --warmup 4 --no-kernel-timing"
run "cfg5 mxfp8 (LN, GELU/dGELU epilogues and attention emit the quantised copies)" A=1 $B --config cfg5
run "cfg5 mxfp8 MMAE_MX_FUSE=0 (separate passes)" MMAE_MX_FUSE=0 $B --config cfg5
run "cfg5 bf16" A=1 $B --config cfg5 --precision bf16
run "cfg5 mxfp8 again" A=1 $B --config cfg5
cat gpurun_out/mx_step3.log
