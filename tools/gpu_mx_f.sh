#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "dw_group or dw" 2>&1 | tail -3
: > gpurun_out/dw_split.log
run() { label="$1"; shift; echo "== $label" >> gpurun_out/dw_split.log; env "$@" > gpurun_out/x.log 2> gpurun_out/x.err; grep "timed region" gpurun_out/x.err | tail -1 | cut -c1-120 >> gpurun_out/dw_split.log; }
B="timeout 600 python bench.py --no-cpu-baseline --steps 15 --warmup 4 --no-kernel-timing"
run "cfg5 bf16, round-filling slices (4)" A=1 $B --config cfg5 --precision bf16
run "cfg5 bf16, MMAE_DW_SPLIT=1 (before)" MMAE_DW_SPLIT=1 $B --config cfg5 --precision bf16
run "cfg5 bf16, MMAE_DW_SPLIT=2" MMAE_DW_SPLIT=2 $B --config cfg5 --precision bf16
run "cfg5 mxfp8, round-filling slices (4)" A=1 $B --config cfg5 --precision mxfp8
run "cfg5 mxfp8, MMAE_DW_SPLIT=1 (before)" MMAE_DW_SPLIT=1 $B --config cfg5 --precision mxfp8
run "cfg3 bf16 (unchanged: 2 slices)" A=1 $B --config cfg3 --precision bf16
cat gpurun_out/dw_split.log
