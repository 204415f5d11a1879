#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; : > gpurun_out/q.log
run() { label="$1"; shift; echo "== $label" >> gpurun_out/q.log; env "$@" > gpurun_out/x.log 2> gpurun_out/x.err; grep "timed region" gpurun_out/x.err | tail -1 | cut -c1-110 >> gpurun_out/q.log; }
B="timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-kernel-timing"
run "default queues" A=1 $B
run "GPU_MAX_HW_QUEUES=8" GPU_MAX_HW_QUEUES=8 $B
run "GPU_MAX_HW_QUEUES=2" GPU_MAX_HW_QUEUES=2 $B
run "GPU_MAX_HW_QUEUES=6" GPU_MAX_HW_QUEUES=6 $B
run "default again" A=1 $B
cat gpurun_out/q.log
