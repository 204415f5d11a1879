#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out
: > gpurun_out/mink.log
run() { label="$1"; shift; echo "== $label" >> gpurun_out/mink.log; env "$@" > gpurun_out/x.log 2> gpurun_out/x.err; grep "timed region" gpurun_out/x.err | tail -1 | cut -c1-120 >> gpurun_out/mink.log; }
B="timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-kernel-timing"
run "default (MMAE_PP_MIN_K=128)" A=1 $B
run "MMAE_PP_MIN_K=257 (K=256 products on the 128x128 kernel)" MMAE_PP_MIN_K=257 $B
run "MMAE_PP_MIN_K=128 MMAE_GEMM_TILE=0 again" A=1 $B
run "MMAE_PP_MIN_K=769" MMAE_PP_MIN_K=769 $B
cat gpurun_out/mink.log
