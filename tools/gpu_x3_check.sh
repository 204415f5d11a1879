#!/bin/bash
# x3 split-K remap: tests + serialized kernel stats of the bench step (average of gemm_f32x3_kernel<true, true>) + the step
export TMPDIR=/tmp
R=$PWD
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "f32x3 or splitk" 2>&1 | tail -2
rm -rf gpurun_out/prof_x3
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_x3 -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary --adapter-streams 0 --wgrad-stream 0 > $R/gpurun_out/prof_x3.log 2>&1)
f=$(find gpurun_out/prof_x3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_x3.csv
rm -rf gpurun_out/prof_x3
grep "gemm_f32x3\|dwgroup" gpurun_out/kernel_stats_x3.csv | cut -d, -f1-4 | cut -c1-140
for i in 1 2; do python bench.py --steps 30 --warmup 8 --no-secondary --no-kernel-timing 2>&1 | grep "timed region" | cut -c1-90; done
