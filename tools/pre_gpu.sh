#!/bin/bash
# run before every gpurun: rebuild, make sure the library loads and the CPU suite is green
set -e
make -C multimae_amd/csrc -j8 2>&1 | grep -E "rror" -A5 || true
python -c "from multimae_amd import _lib; _lib.load(); print('lib ok')"
timeout 900 python -m pytest tests -x -q -m "not gpu" 2>&1 | tail -2
