#!/bin/bash
# GPU visit B: why is the one-call-per-stack host path slower on the GPU?  stream-configuration matrix + kernel traces.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
: > gpurun_out/summary.txt
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 3"
run() { # label, env..., -- args
  label=$1; shift
  ( env "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //')" >> gpurun_out/summary.txt
}
for comp in 1 0; do for as in 1 0; do for ws in 1 0; do
  run "stack_composite=$comp adapter_streams=$as wgrad_stream=$ws" MMAE_STACK_COMPOSITE=$comp timeout 300 $B --adapter-streams $as --wgrad-stream $ws
done; done; done
run "stack=1 adapter_composite=0 (encoder one call, adapters per-block) as=1 ws=1" MMAE_ADAPTER_COMPOSITE=0 timeout 300 $B
run "stack=1 adapter_composite=0 as=0 ws=1" MMAE_ADAPTER_COMPOSITE=0 timeout 300 $B --adapter-streams 0
run "graph=1 stack=1" timeout 300 $B --graph 1
run "graph=1 stack=1 as=0" timeout 300 $B --graph 1 --adapter-streams 0
run "GPU_MAX_HW_QUEUES=8 stack=1 as=1 ws=1" GPU_MAX_HW_QUEUES=8 timeout 300 $B
run "GPU_MAX_HW_QUEUES=2 stack=1 as=1 ws=1" GPU_MAX_HW_QUEUES=2 timeout 300 $B
for mode in new old; do
  E=1; [ $mode = old ] && E=0
  rm -rf gpurun_out/prof_$mode
  (cd /tmp && MMAE_STACK_COMPOSITE=$E timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$mode -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_$mode.log 2>&1)
  f=$(find gpurun_out/prof_$mode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_$mode.csv
  t=$(find gpurun_out/prof_$mode -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/trace_union.py gpurun_out/prof_$mode > gpurun_out/trace_union_$mode.txt 2>&1
  [ -n "$t" ] && gzip -c "$t" > gpurun_out/kernel_trace_$mode.csv.gz
  rm -rf gpurun_out/prof_$mode
  grep "timed region" gpurun_out/prof_$mode.log >> gpurun_out/summary.txt
done
cat gpurun_out/summary.txt
