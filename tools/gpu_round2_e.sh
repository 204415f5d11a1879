#!/bin/bash
# GPU visit E: patch-domain losses (tests + bench A/B).
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
: > gpurun_out/summary.txt
echo "== pytest -m gpu (all)" >> gpurun_out/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
tail -25 gpurun_out/pytest_gpu.log >> gpurun_out/summary.txt
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5"
run() { label=$1; shift; ( env "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //')" >> gpurun_out/summary.txt; }
run "patch_loss=1 (default)" timeout 300 $B
run "patch_loss=0" MMAE_PATCH_LOSS=0 timeout 300 $B
run "patch_loss=1 serialized" timeout 300 $B --adapter-streams 0 --wgrad-stream 0
run "patch_loss=0 serialized" MMAE_PATCH_LOSS=0 timeout 300 $B --adapter-streams 0 --wgrad-stream 0
run "patch_loss=1 (default) again" timeout 300 $B
echo "== smoke" >> gpurun_out/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200 >> gpurun_out/summary.txt
cat gpurun_out/summary.txt
