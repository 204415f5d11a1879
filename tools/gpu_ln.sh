#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "layernorm or ln or block or stack" 2>&1 | tail -3
: > gpurun_out/ln.log
run() { label="$1"; shift; echo "== $label" >> gpurun_out/ln.log; env "$@" > gpurun_out/x.log 2> gpurun_out/x.err; grep "timed region" gpurun_out/x.err | tail -1 | cut -c1-120 >> gpurun_out/ln.log; }
B="timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-kernel-timing"
run "cfg3 resident LN-bwd grid" A=1 $B
run "cfg3 MMAE_LN_BWD_RESIDENT=0 (1024 workgroups)" MMAE_LN_BWD_RESIDENT=0 $B
run "cfg3 resident LN-bwd grid, again" A=1 $B
run "cfg3 MMAE_LN_BWD_RESIDENT=0, again" MMAE_LN_BWD_RESIDENT=0 $B
run "cfg5 bf16 resident" A=1 $B --config cfg5
run "cfg5 bf16 MMAE_LN_BWD_RESIDENT=0" MMAE_LN_BWD_RESIDENT=0 $B --config cfg5
cat gpurun_out/ln.log
