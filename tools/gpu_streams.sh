#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; : > gpurun_out/streams.log
run() { label="$1"; shift; echo "== $label" >> gpurun_out/streams.log; "$@" > gpurun_out/x.log 2> gpurun_out/x.err; grep "timed region" gpurun_out/x.err | tail -1 | cut -c1-110 >> gpurun_out/streams.log; }
B="timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-kernel-timing"
run "adapter streams on, wgrad stream on (default)" $B
run "adapter streams on, wgrad stream off" $B --wgrad-stream 0
run "adapter streams off, wgrad stream on" $B --adapter-streams 0
run "both off (one stream)" $B --adapter-streams 0 --wgrad-stream 0
cat gpurun_out/streams.log
