#!/bin/bash
# GPU visit F: grouped-dW tile order / slice count experiments (serialized kernel time is what counts) + kernel test.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
: > gpurun_out/summary.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "dw_group" -p no:cacheprovider 2>&1 | tail -2 >> gpurun_out/summary.txt
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5"
run() { label=$1; shift; ( env "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-60)" >> gpurun_out/summary.txt; }
for v in "MMAE_DW_XCD=1" "MMAE_DW_XCD=0" "MMAE_DW_SPLIT=4" "MMAE_DW_SPLIT=7" "MMAE_DW_GROUP=0"; do
  run "$v default-streams" $v timeout 300 $B
  run "$v serialized" $v timeout 300 $B --adapter-streams 0 --wgrad-stream 0
done
for v in "MMAE_DW_XCD=1" "MMAE_DW_XCD=0" "MMAE_DW_SPLIT=7"; do
  rm -rf gpurun_out/prof_x
  (cd /tmp && env $v timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_x -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --adapter-streams 0 --wgrad-stream 0 > $R/gpurun_out/prof_x.log 2>&1)
  f=$(find gpurun_out/prof_x -name "*kernel_stats.csv" | head -1)
  echo "== $v kernel stats" >> gpurun_out/summary.txt
  [ -n "$f" ] && grep -E "dwgroup|dw_group_reduce" "$f" | cut -d, -f1-5 | cut -c1-160 >> gpurun_out/summary.txt
  rm -rf gpurun_out/prof_x
done
cat gpurun_out/summary.txt
