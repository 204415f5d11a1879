#!/bin/bash
# GPU visit D: full parity suite, serialized kernel stats + PMC traffic with the grouped weight gradients.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
: > gpurun_out/summary.txt
echo "== pytest -m gpu (all)" >> gpurun_out/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
tail -25 gpurun_out/pytest_gpu.log >> gpurun_out/summary.txt
PROF="--steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --adapter-streams 0 --wgrad-stream 0"
rm -rf gpurun_out/prof_serialized
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_serialized -o p --output-format csv -- python $R/bench.py $PROF > $R/gpurun_out/prof_serialized.log 2>&1)
f=$(find gpurun_out/prof_serialized -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_serialized.csv
rm -rf gpurun_out/prof_serialized
grep "timed region" gpurun_out/prof_serialized.log >> gpurun_out/summary.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  (cd /tmp && timeout 600 rocprofv3 --pmc $c -d $R/gpurun_out/pmc_$c -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --adapter-streams 0 --wgrad-stream 0 > $R/gpurun_out/pmc_$c.log 2>&1)
done
python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_traffic.json >> gpurun_out/summary.txt 2>&1
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
cat gpurun_out/summary.txt
