#!/bin/bash
# full GPU suite after the ABI bump + MX mode, smoke, default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; grep "timed region" gpurun_out/bench.err | cut -c1-160
