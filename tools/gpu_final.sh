#!/bin/bash
# Round-end GPU visit: parity tests, smoke, the default bench line, cfg2 / cfg5 lines, same-box A/B of this round's switches,
# rocprofv3 kernel stats of the default and of the serialized (one stream) step, the PMC passes for HBM traffic and MFMA
# occupancy, the per-GEMM table of an encoder block and the encoder step.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
: > gpurun_out/summary.txt
echo "== pytest -m gpu" >> gpurun_out/summary.txt
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
tail -4 gpurun_out/pytest_gpu.log >> gpurun_out/summary.txt
echo "== smoke" >> gpurun_out/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-220 >> gpurun_out/summary.txt
echo "== bench (default flags)" >> gpurun_out/summary.txt
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err
grep "timed region\|cpu_baseline:" gpurun_out/bench.err >> gpurun_out/summary.txt
tail -1 gpurun_out/bench.log >> gpurun_out/summary.txt
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5"
run() { label=$1; shift; ( env "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-120)" >> gpurun_out/summary.txt; }
echo "== same-box A/B (cfg3, B=256)" >> gpurun_out/summary.txt
run "all switches on (default)" timeout 300 $B
run "MMAE_PP_FL=0 (run-time epilogue dispatch)" MMAE_PP_FL=0 timeout 300 $B
run "MMAE_GEMM_T10_CS=0 (256-row tiles for dGELU+colsum)" MMAE_GEMM_T10_CS=0 timeout 300 $B
run "MMAE_DW_GROUP=0 (one split-K launch per weight gradient)" MMAE_DW_GROUP=0 timeout 300 $B
run "MMAE_PATCH_LOSS=0 (image-domain losses)" MMAE_PATCH_LOSS=0 timeout 300 $B
run "MMAE_STACK_COMPOSITE=0 (round-1 host path: one call per block)" MMAE_STACK_COMPOSITE=0 timeout 300 $B
run "all of the above off (round-1 configuration)" MMAE_PP_FL=0 MMAE_GEMM_T10_CS=0 MMAE_DW_GROUP=0 MMAE_PATCH_LOSS=0 MMAE_STACK_COMPOSITE=0 timeout 300 $B
run "all switches on (default), again" timeout 300 $B
run "serialized (one stream)" timeout 300 $B --adapter-streams 0 --wgrad-stream 0
run "hipGraph replay" timeout 300 $B --graph 1
echo "== other configs" >> gpurun_out/summary.txt
run "cfg2 (RGB only) B=256" timeout 300 $B --config cfg2
run "cfg5 geometry (ViT-L, bf16) B=128" timeout 600 $B --config cfg5 --precision bf16 --steps 10 --warmup 3
run "cfg5 geometry (ViT-L, --precision mxfp8) B=128" timeout 600 $B --config cfg5 --precision mxfp8 --steps 10 --warmup 3
run "cfg3, data-parallel path over RCCL at world size 1 (--force-dist 1)" timeout 300 $B --force-dist 1
run "cfg3 B=128" timeout 300 $B --batch 128
run "cfg3 B=512" timeout 300 $B --batch 512 --steps 10 --warmup 3
PROF="--steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing"
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
  (cd /tmp && timeout 600 rocprofv3 --pmc $c -d $R/gpurun_out/pmc_$c -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --adapter-streams 0 --wgrad-stream 0 > $R/gpurun_out/pmc_$c.log 2>&1)
done
python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_traffic.json >> gpurun_out/summary.txt 2>&1
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
echo "== per-GEMM table of an encoder block (rocprofv3 kernel trace, each product alone)" >> gpurun_out/summary.txt
rm -rf gpurun_out/encg
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/encg -o p --output-format csv -- python $R/tools/encoder_gemms.py > $R/gpurun_out/encg.log 2>&1)
python tools/encoder_gemms.py --parse gpurun_out/encg > gpurun_out/encoder_gemms.txt 2>&1
cat gpurun_out/encoder_gemms.txt >> gpurun_out/summary.txt
rm -rf gpurun_out/encg
echo "== MFMA occupancy PMC (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE) of the same products" >> gpurun_out/summary.txt
rm -rf gpurun_out/pmc_mfma
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_mfma -o p --output-format csv -- python $R/tools/encoder_gemms.py > $R/gpurun_out/pmc_mfma.log 2>&1)
python tools/pmc_mfma.py gpurun_out/pmc_mfma > gpurun_out/pmc_mfma.txt 2>&1
cat gpurun_out/pmc_mfma.txt >> gpurun_out/summary.txt
rm -rf gpurun_out/pmc_mfma
echo "== encoder step" >> gpurun_out/summary.txt
timeout 300 python tools/encoder_step.py > gpurun_out/encoder_step.json 2> gpurun_out/encoder_step.err
cat gpurun_out/encoder_step.json >> gpurun_out/summary.txt
cat gpurun_out/summary.txt
