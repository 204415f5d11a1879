#!/bin/bash
# GPU visit: parity tests, then bench eager vs hipGraph, then rocprofv3 kernel stats of the default (graph) bench.
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/summary.txt
echo "== pytest -m gpu ${PYTEST_ARGS}" >> gpurun_out/summary.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1
tail -30 gpurun_out/pytest_gpu.log >> gpurun_out/summary.txt
for mode in ${MODES:-0 1}; do
  echo "== bench --graph $mode" >> gpurun_out/summary.txt
  timeout 600 python bench.py --steps 20 --warmup 5 --graph $mode ${BENCH_ARGS} > gpurun_out/bench_g$mode.log 2> gpurun_out/bench_g$mode.err
  grep -E "timed region|captured|Error|error" gpurun_out/bench_g$mode.err | tail -5 >> gpurun_out/summary.txt
  tail -3 gpurun_out/bench_g$mode.err >> gpurun_out/summary.txt
  tail -1 gpurun_out/bench_g$mode.log >> gpurun_out/summary.txt
done
if [ -z "$NOPROF" ]; then
echo "== rocprofv3 kernel stats (default bench)" >> gpurun_out/summary.txt
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof -o r1 --output-format csv -- python $OLDPWD/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing ${BENCH_ARGS} > $OLDPWD/gpurun_out/prof.log 2>&1)
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/r1_kernel_stats.csv; head -25 "$f" | cut -c1-160 >> gpurun_out/summary.txt; else tail -20 gpurun_out/prof.log >> gpurun_out/summary.txt; fi
tail -3 gpurun_out/prof.log >> gpurun_out/summary.txt
fi
cat gpurun_out/summary.txt
