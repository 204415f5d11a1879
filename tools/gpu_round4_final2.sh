#!/bin/bash
# Round-4 closing GPU visit (after the fp16-storage adapter became the default): parity tests, smoke, the default bench line (+ secondary),
# cfg2 / cfg5 lines, the data-parallel path at world size 1, the drop-in loop, the 'f16' adapter mode beside the default, rocprofv3 kernel
# stats (default and serialized), HBM-traffic PMC passes, the encoder step.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
: > gpurun_out/summary.txt
if [ "$1" != "notests" ]; then
echo "== pytest -m gpu" >> gpurun_out/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
tail -6 gpurun_out/pytest_gpu.log >> gpurun_out/summary.txt
echo "== smoke" >> gpurun_out/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-220 >> gpurun_out/summary.txt
fi
if [ "$1" = "testsonly" ]; then cat gpurun_out/summary.txt; exit 0; fi
echo "== bench (default flags)" >> gpurun_out/summary.txt
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err
grep "timed region\|cpu_baseline:" gpurun_out/bench.err >> gpurun_out/summary.txt
grep "^{" gpurun_out/bench.log | tail -1 > gpurun_out/bench_cfg3.json
cat gpurun_out/bench_cfg3.json >> gpurun_out/summary.txt
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5"
run() { label=$1; shift; ( "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-120)" >> gpurun_out/summary.txt; }
echo "== other configurations" >> gpurun_out/summary.txt
run "cfg3 default, again" timeout 300 $B
run "cfg3, fp32 adapter as f32 tensors with fp16 operands (--fp32-adapter-gemm f16, the default before the fp16-storage mode)" timeout 300 $B --fp32-adapter-gemm f16
run "cfg3 serialized (one stream)" timeout 300 $B --adapter-streams 0 --wgrad-stream 0
run "cfg2 (RGB only) B=256" timeout 300 $B --config cfg2
run "cfg5 geometry (ViT-L, bf16) B=128" timeout 600 $B --config cfg5 --precision bf16 --steps 10 --warmup 3
timeout 600 python bench.py --no-cpu-baseline --no-secondary --config cfg5 --precision mxfp8 --steps 10 --warmup 3 > gpurun_out/bench_cfg5_mxfp8.log 2> gpurun_out/x.err
grep "^{" gpurun_out/bench_cfg5_mxfp8.log | tail -1 > gpurun_out/bench_cfg5_mxfp8.json
echo "cfg5 (ViT-L, mxfp8) B=128: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-120)" >> gpurun_out/summary.txt
timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5 --force-dist 1 > gpurun_out/dist_world1.log 2> gpurun_out/x.err
echo "cfg3, data-parallel path over RCCL at world size 1, collectives issued (--force-dist 1): $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-120)" >> gpurun_out/summary.txt
grep "^{" gpurun_out/dist_world1.log | tail -1 > gpurun_out/dist_world1_rccl.json
run "cfg3, the reference's loop body through the drop-in boundary (--dropin-ddp 1)" timeout 300 $B --dropin-ddp 1
PROF="--steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary"
for mode in default serialized; do
  EXTRA=""; [ $mode = serialized ] && EXTRA="--adapter-streams 0 --wgrad-stream 0"
  rm -rf gpurun_out/prof_$mode
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$mode -o p --output-format csv -- python $R/bench.py $PROF $EXTRA > $R/gpurun_out/prof_$mode.log 2>&1)
  f=$(find gpurun_out/prof_$mode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_$mode.csv
  python tools/trace_union.py gpurun_out/prof_$mode > gpurun_out/trace_union_$mode.txt 2>&1
  rm -rf gpurun_out/prof_$mode
  grep "timed region" gpurun_out/prof_$mode.log | cut -c1-160 >> gpurun_out/summary.txt
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  (cd /tmp && timeout 600 rocprofv3 --pmc $c -d $R/gpurun_out/pmc_$c -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-secondary --adapter-streams 0 --wgrad-stream 0 > $R/gpurun_out/pmc_$c.log 2>&1)
done
python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_traffic.json >> gpurun_out/summary.txt 2>&1
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
echo "== encoder step" >> gpurun_out/summary.txt
timeout 300 python tools/encoder_step.py > gpurun_out/encoder_step.json 2> gpurun_out/encoder_step.err
cat gpurun_out/encoder_step.json >> gpurun_out/summary.txt
cat gpurun_out/summary.txt
