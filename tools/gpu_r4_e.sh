#!/bin/bash
# Round-4 visit E: chip-sharing A/Bs (adapter grids side by side; encoder dX chain beside its weight gradients), one box.
mkdir -p gpurun_out
export TMPDIR=/tmp
S=gpurun_out/r4e_summary.txt
: > $S
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5"
run() { label=$1; shift; ( "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-60)" >> $S; tail -2 gpurun_out/x.err | grep -i "error\|Traceback" >> $S; }
run "default" timeout 300 $B
run "adapter-cu-share 128" timeout 300 $B --adapter-cu-share 128
run "adapter-cu-share 64" timeout 300 $B --adapter-cu-share 64
run "adapter-cu-share 160" timeout 300 $B --adapter-cu-share 160
run "enc-bwd-side-cus 108" timeout 300 $B --enc-bwd-side-cus 108
run "enc-bwd-side-cus 54" timeout 300 $B --enc-bwd-side-cus 54
run "default (again)" timeout 300 $B
run "adapter-cu-share 128 + enc-bwd-side-cus 108" timeout 300 $B --adapter-cu-share 128 --enc-bwd-side-cus 108
timeout 300 python tools/encoder_step.py > gpurun_out/r4e_encoder_step.json 2> gpurun_out/x.err
echo "encoder step: $(cat gpurun_out/r4e_encoder_step.json)" >> $S
cat $S
