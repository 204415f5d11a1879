#!/bin/bash
# One GPU-box visit: parity tests, smoke, GEMM micro-bench, bench line.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu" > gpurun_out/summary.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
tail -60 gpurun_out/pytest_gpu.log >> gpurun_out/summary.txt
echo "== smoke" >> gpurun_out/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -5 gpurun_out/smoke.log >> gpurun_out/summary.txt
echo "== gemm bench" >> gpurun_out/summary.txt
timeout 300 python tools/gemm_bench.py > gpurun_out/gemm_bench.log 2>&1; cat gpurun_out/gemm_bench.log >> gpurun_out/summary.txt
echo "== bench" >> gpurun_out/summary.txt
timeout 600 python bench.py --steps 5 --warmup 2 ${BENCH_ARGS} > gpurun_out/bench.log 2>&1; tail -5 gpurun_out/bench.log >> gpurun_out/summary.txt
cat gpurun_out/summary.txt
