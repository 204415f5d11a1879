#!/bin/bash
# One GPU-box visit: parity tests, smoke, GEMM micro-bench, bench line, rocprofv3 kernel stats.
# Everything lands in gpurun_out/.  Stages can be selected: STAGES="test smoke gemm bench prof"
STAGES=${STAGES:-"test smoke gemm bench prof"}
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/summary.txt
for s in $STAGES; do
case $s in
test)
  echo "== pytest -m gpu" >> gpurun_out/summary.txt
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1
  tail -40 gpurun_out/pytest_gpu.log >> gpurun_out/summary.txt ;;
smoke)
  echo "== smoke" >> gpurun_out/summary.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -5 gpurun_out/smoke.log >> gpurun_out/summary.txt ;;
gemm)
  echo "== gemm bench" >> gpurun_out/summary.txt
  timeout 300 python tools/gemm_bench.py > gpurun_out/gemm_bench.log 2>&1; cat gpurun_out/gemm_bench.log >> gpurun_out/summary.txt ;;
bench)
  echo "== bench" >> gpurun_out/summary.txt
  timeout 600 python bench.py --steps 10 --warmup 3 ${BENCH_ARGS} > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -12 gpurun_out/bench.err >> gpurun_out/summary.txt; tail -2 gpurun_out/bench.log >> gpurun_out/summary.txt ;;
prof)
  echo "== rocprofv3 kernel stats" >> gpurun_out/summary.txt
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof -o r1 --output-format csv -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing ${BENCH_ARGS} > $OLDPWD/gpurun_out/prof.log 2>&1)
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" gpurun_out/r1_kernel_stats.csv; head -40 "$f" >> gpurun_out/summary.txt; else tail -20 gpurun_out/prof.log >> gpurun_out/summary.txt; fi
  # keep the merge small: drop the raw per-dispatch trace
  find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete ;;
esac
done
cat gpurun_out/summary.txt
