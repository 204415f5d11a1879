#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
: > gpurun_out/summary.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8 >> gpurun_out/summary.txt
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5"
run() { label=$1; shift; ( env "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-60)" >> gpurun_out/summary.txt; }
run "default-streams" timeout 300 $B
run "serialized" timeout 300 $B --adapter-streams 0 --wgrad-stream 0
run "default-streams again" timeout 300 $B
rm -rf gpurun_out/encg
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/encg -o p --output-format csv -- python $R/tools/encoder_gemms.py > $R/gpurun_out/encg.log 2>&1)
python tools/encoder_gemms.py --parse gpurun_out/encg >> gpurun_out/summary.txt 2>&1
rm -rf gpurun_out/encg
timeout 300 python tools/encoder_step.py >> gpurun_out/summary.txt 2> gpurun_out/encoder_step.err
cat gpurun_out/summary.txt
