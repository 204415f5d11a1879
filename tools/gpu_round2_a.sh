#!/bin/bash
# GPU visit A (round 2): full parity suite on the composite paths, bench default + host-enqueue check, old-path A/B.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
: > gpurun_out/summary.txt
echo "== pytest -m gpu" >> gpurun_out/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
tail -30 gpurun_out/pytest_gpu.log >> gpurun_out/summary.txt
echo "== smoke" >> gpurun_out/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> gpurun_out/summary.txt
echo "== bench (default flags)" >> gpurun_out/summary.txt
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err
grep "timed region\|cpu_baseline" gpurun_out/bench.err >> gpurun_out/summary.txt
tail -1 gpurun_out/bench.log >> gpurun_out/summary.txt
echo "== bench, per-block composites (round-1 host path)" >> gpurun_out/summary.txt
MMAE_STACK_COMPOSITE=0 timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_oldhost.log 2> gpurun_out/bench_oldhost.err
grep "timed region" gpurun_out/bench_oldhost.err >> gpurun_out/summary.txt
echo "== bench cfg2" >> gpurun_out/summary.txt
timeout 600 python bench.py --config cfg2 --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_cfg2.log 2> gpurun_out/bench_cfg2.err
grep "timed region" gpurun_out/bench_cfg2.err >> gpurun_out/summary.txt
echo "== encoder step" >> gpurun_out/summary.txt
timeout 300 python tools/encoder_step.py > gpurun_out/encoder_step.json 2> gpurun_out/encoder_step.err
cat gpurun_out/encoder_step.json >> gpurun_out/summary.txt
cat gpurun_out/summary.txt
