#!/bin/bash
# the data-parallel path on ONE GPU over RCCL (world size 1): process-group init, bucketed all-reduce launched during backward
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/dist1.log
for mode in "--force-dist 1" ""; do
  echo "== bench.py $mode" >> gpurun_out/dist1.log
  timeout 600 python bench.py $mode --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 > gpurun_out/x.log 2> gpurun_out/x.err
  echo "rc=$?" >> gpurun_out/dist1.log
  grep "timed region\|Error\|error" gpurun_out/x.err | tail -3 | cut -c1-200 >> gpurun_out/dist1.log
done
cat gpurun_out/dist1.log
tail -5 gpurun_out/x.err | cut -c1-200
