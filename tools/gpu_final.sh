#!/bin/bash
# Round-end GPU visit: parity tests, smoke, the default bench line, rocprofv3 kernel stats of the default and of the serialized
# (one stream) step, and the two PMC passes for HBM traffic.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
: > gpurun_out/summary.txt
echo "== pytest -m gpu" >> gpurun_out/summary.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
tail -4 gpurun_out/pytest_gpu.log >> gpurun_out/summary.txt
echo "== smoke" >> gpurun_out/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/summary.txt
echo "== bench (default flags)" >> gpurun_out/summary.txt
timeout 600 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err
grep "timed region" gpurun_out/bench.err >> gpurun_out/summary.txt
tail -1 gpurun_out/bench.log >> gpurun_out/summary.txt
PROF="--steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing"
for mode in default serialized; do
  EXTRA=""; [ $mode = serialized ] && EXTRA="--adapter-streams 0 --wgrad-stream 0"
  rm -rf gpurun_out/prof_$mode
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$mode -o p --output-format csv -- python $R/bench.py $PROF $EXTRA > $R/gpurun_out/prof_$mode.log 2>&1)
  f=$(find gpurun_out/prof_$mode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_$mode.csv
  find gpurun_out/prof_$mode -name "*kernel_trace.csv" -size +8M -delete
  grep "timed region" gpurun_out/prof_$mode.log >> gpurun_out/summary.txt
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  (cd /tmp && timeout 600 rocprofv3 --pmc $c -d $R/gpurun_out/pmc_$c -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --adapter-streams 0 --wgrad-stream 0 > $R/gpurun_out/pmc_$c.log 2>&1)
done
python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_traffic.json >> gpurun_out/summary.txt 2>&1
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -size +8M -delete
cat gpurun_out/summary.txt
