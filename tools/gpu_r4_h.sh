#!/bin/bash
# round 4, visit h: the whole GPU suite with the fp16-storage adapter as the default + a serialized kernel profile
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
S=gpurun_out/r4h_summary.txt
: > $S
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r4h_pytest.log 2>&1
tail -30 gpurun_out/r4h_pytest.log | grep -E "passed|failed|FAILED|Error" | head -20 >> $S
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200 >> $S
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5"
run() { label=$1; shift; ( "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-60)" >> $S; tail -3 gpurun_out/x.err | grep -i "error\|Traceback" >> $S; }
run "default (h16)" timeout 300 $B
run "cfg5 bf16" timeout 600 $B --config cfg5 --precision bf16 --steps 10 --warmup 3
run "dropin-ddp" timeout 300 $B --dropin-ddp 1
run "force-dist" timeout 300 $B --force-dist 1
rm -rf gpurun_out/prof_serialized
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_serialized -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary --adapter-streams 0 --wgrad-stream 0 > $R/gpurun_out/prof_serialized.log 2>&1)
f=$(find gpurun_out/prof_serialized -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4h_kernel_stats_serialized.csv
rm -rf gpurun_out/prof_serialized
cat $S
