#!/bin/bash
# Round 5: ONE parameterised collection script (replaces the per-visit gpu_r4_*.sh files).  Usage on the GPU box:
#   gpurun -- 'bash tools/gpu_r5.sh <visit> [...]'      visits: duo | baseline | tests | final | ...
# Everything lands under gpurun_out/r5_<visit>_*; what is worth keeping is copied to profiles/ by hand (profiles/INDEX.md).
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
V=${1:-baseline}; shift
S=gpurun_out/r5_${V}_summary.txt
: > $S
EXP=$R/multimae_amd/libmmae_hip_exp.so
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5"
# run <label> <cmd...>: one bench run, its step time into the summary
run() { label=$1; shift; ( "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-70)" >> $S; grep -i "error\|Traceback" gpurun_out/x.err | tail -3 >> $S; }
# table <tool.py> <label> [tool args]: a per-product GEMM table under rocprofv3 --kernel-trace
table() { tool=$1; label=$2; shift 2; rm -rf gpurun_out/tbl; (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/tbl -o p --output-format csv -- python $R/tools/$tool "$@" > $R/gpurun_out/tbl.log 2>&1); echo "== $label" >> $S; python tools/$tool --parse gpurun_out/tbl "$@" >> $S 2>&1; rm -rf gpurun_out/tbl; }
# kstats <tag> [bench args]: serialized kernel stats of the bench command -> gpurun_out/r5_<visit>_kernel_stats_<tag>.csv
kstats() { tag=$1; shift; rm -rf gpurun_out/prof; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary --adapter-streams 0 --wgrad-stream 0 "$@" > $R/gpurun_out/prof.log 2>&1); f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r5_${V}_kernel_stats_$tag.csv; rm -rf gpurun_out/prof; }

case $V in
duo)   # the two-workgroups-per-CU GEMM (tile 11, experiments library) on the output adapters' short-K products
  run "production library, defaults" timeout 300 $B
  [ -f $EXP ] || { echo "the experiments library did not travel: comment its line out of .gpurunignore (and run make -C multimae_amd/csrc exp)" >> $S; exit 1; }
  export MMAE_LIB=$EXP
  run "exp library, defaults" timeout 300 $B
  MMAE_DUO_MAX_K=256 run "exp, duo for K <= 256" timeout 300 $B
  MMAE_DUO_MAX_K=1024 run "exp, duo for K <= 1024" timeout 300 $B
  run "exp library, defaults again" timeout 300 $B
  table decoder_gemms.py "decoder GEMMs, tiles 9 10 11 12" 9 10 11 12
  ;;
probe)   # HBM ceilings by direction, the encoder table and the serialized kernel statistics of this box
  python tools/hbm_probe.py >> $S 2>&1
  table encoder_gemms.py "encoder GEMMs"
  kstats base
  python - >> $S <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/r5_probe_kernel_stats_base.csv')))
print('== serialized ms/step', round(sum(float(r['TotalDurationNs']) for r in rows) / 8e6, 3))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:45]:
    print(f"{float(r['TotalDurationNs']) / 8e6:7.3f} ms/step  {int(r['Calls']) / 8:6.1f} calls  {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:150]}")
PY
  ;;
newtests)   # the tests added this round + the default bench line
  timeout 1500 python -m pytest tests/test_curves_gpu.py tests/test_dropin_loop_gpu.py tests/test_h16_gpu.py -x -q 2>&1 | tail -25 >> $S
  for f in curve_cfg3_bf16_h16.json curve_cfg3_bf16_f16.json curve_cfg3_bf16_x3.json adapter_modes_train_cfg3.json mxfp8_trains_cfg5.json dropin_fast_loop.json; do [ -f gpurun_out/$f ] && cp gpurun_out/$f gpurun_out/r5_$f; done
  run "production library, defaults" timeout 300 $B
  ;;
pmcdec)   # HBM bytes per launch of the decoder / encoder products (tile 10 or the planner's choice), FETCH and WRITE passes
  for c in FETCH_SIZE WRITE_SIZE; do rm -rf gpurun_out/pmc_$c; (cd /tmp && timeout 300 rocprofv3 --pmc $c -d $R/gpurun_out/pmc_$c -o p --output-format csv -- python $R/tools/decoder_gemms.py 10 > $R/gpurun_out/pmc_$c.log 2>&1); done
  echo "== decoder products, tile 10: HBM bytes per launch" >> $S
  python tools/pmc_cases.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE 8 "fwd qkv" "fwd proj" "fwd fc1" "fwd fc2" "dx fc2" "dx fc1" "dx proj" "dx qkv" >> $S 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do rm -rf gpurun_out/pmc_$c; (cd /tmp && timeout 300 rocprofv3 --pmc $c -d $R/gpurun_out/pmc_$c -o p --output-format csv -- python $R/tools/encoder_gemms.py > $R/gpurun_out/pmc_$c.log 2>&1); done
  echo "== encoder products: HBM bytes per launch" >> $S
  python tools/pmc_cases.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE 12 "fwd qkv" "fwd proj" "fwd fc1" "fwd fc2" "dx fc2" "dx fc1" "dx proj" "dx qkv" "dw fc2" "dw fc1" "dw proj" "dw qkv" >> $S 2>&1
  rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
  ;;
dephase)   # odd workgroups of the two-stream-epilogue products start ~4 us late (experiments library)
  [ -f $EXP ] || { echo "the experiments library did not travel: comment its line out of .gpurunignore (and run make -C multimae_amd/csrc exp)" >> $S; exit 1; }
  export MMAE_LIB=$EXP
  run "exp library, defaults" timeout 300 $B
  MMAE_PP_DEPHASE=1 MMAE_PP_DEPHASE_SEL=1 run "exp, dephase 1 on GELU / dGELU epilogues, K >= 512" timeout 300 $B
  run "exp library, defaults again" timeout 300 $B
  MMAE_PP_DEPHASE=1 MMAE_PP_DEPHASE_SEL=1 run "exp, dephase 1 sel again" timeout 300 $B
  MMAE_PP_DEPHASE=1 MMAE_PP_DEPHASE_SEL=1 table encoder_gemms.py "encoder GEMMs, dephase 1 on the two-stream epilogues"
  table encoder_gemms.py "encoder GEMMs, defaults"
  ;;
quick)   # the adapter-level parity tests + the default line (after a change inside mmae_adapter_fwd / _bwd)
  timeout 1500 python -m pytest tests/test_parity_geometry_gpu.py tests/test_kernels_gpu.py tests/test_h16_gpu.py tests/test_model_gpu.py -x -q -k "per_tensor or mask_token or decoder or adapter or build or golden or h16 or smoke or mini" 2>&1 | tail -8 >> $S
  run "production library, defaults" timeout 300 $B
  run "production library, defaults again" timeout 300 $B
  ;;
firstwrite)   # gradient store-on-first-write: its tests, then the A/B inside one visit
  timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "first_write or mini_model or training_loop or graph" 2>&1 | tail -5 >> $S
  run "defaults (first write stores)" timeout 300 $B
  MMAE_FIRST_WRITE=0 run "first write accumulates (MMAE_FIRST_WRITE=0)" timeout 300 $B
  run "defaults again" timeout 300 $B
  MMAE_FIRST_WRITE=0 run "accumulate again" timeout 300 $B
  ;;
tests)   # the whole GPU suite + smoke
  echo "== pytest -m gpu" >> $S
  timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r5_pytest_gpu.log 2>&1
  tail -8 gpurun_out/r5_pytest_gpu.log >> $S
  echo "== smoke" >> $S
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-220 >> $S
  for f in curve_cfg3_bf16_h16.json adapter_modes_train_cfg3.json mxfp8_trains_cfg5.json dropin_fast_loop.json; do [ -f gpurun_out/$f ] && cp gpurun_out/$f gpurun_out/r5_$f; done
  ;;
final)   # the round's measurement visit: default line (+ secondaries), other configurations, kernel statistics, PMC traffic, tables, encoder step
  echo "== bench (default flags)" >> $S
  timeout 1200 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err
  grep "timed region\|cpu_baseline" gpurun_out/bench.err | cut -c1-200 >> $S
  grep "^{" gpurun_out/bench.log | tail -1 > gpurun_out/r5_bench_cfg3.json
  cat gpurun_out/r5_bench_cfg3.json >> $S
  echo "== other configurations" >> $S
  run "cfg3 default, again" timeout 300 $B
  MMAE_GELU_GRAD_AUX=0 run "cfg3, the MLP pair with the pre-activation stored and GELU' re-evaluated by the backward (MMAE_GELU_GRAD_AUX=0)" timeout 300 $B
  run "cfg3 default, a third time" timeout 300 $B
  run "cfg3, fp32 adapter as f32 tensors with fp16 operands (--fp32-adapter-gemm f16)" timeout 300 $B --fp32-adapter-gemm f16
  run "cfg3, fp32 adapter with split-bf16 operands (--fp32-adapter-gemm x3)" timeout 300 $B --fp32-adapter-gemm x3
  run "cfg3 serialized (one stream)" timeout 300 $B --adapter-streams 0 --wgrad-stream 0
  run "cfg2 (RGB only) B=256" timeout 300 $B --config cfg2
  run "cfg5 geometry (ViT-L, bf16) B=128" timeout 600 $B --config cfg5 --precision bf16 --steps 10 --warmup 3
  timeout 600 python bench.py --no-cpu-baseline --no-secondary --config cfg5 --precision mxfp8 --steps 10 --warmup 3 > gpurun_out/bench_cfg5_mxfp8.log 2> gpurun_out/x.err
  grep "^{" gpurun_out/bench_cfg5_mxfp8.log | tail -1 > gpurun_out/r5_bench_cfg5_mxfp8.json
  echo "cfg5 (ViT-L, mxfp8) B=128: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-120)" >> $S
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5 --force-dist 1 > gpurun_out/dist_world1.log 2> gpurun_out/x.err
  echo "cfg3, data-parallel path over RCCL at world size 1, collectives issued (--force-dist 1): $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-120)" >> $S
  grep "^{" gpurun_out/dist_world1.log | tail -1 > gpurun_out/r5_dist_world1_rccl.json
  run "cfg3, the reference's loop body through the plain drop-in seam (--dropin-ddp 1: torch DDP + GradScaler + per-parameter norm)" timeout 300 $B --dropin-ddp 1
  PROF="--steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary"
  for mode in default serialized; do
    EXTRA=""; [ $mode = serialized ] && EXTRA="--adapter-streams 0 --wgrad-stream 0"
    rm -rf gpurun_out/prof_$mode
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$mode -o p --output-format csv -- python $R/bench.py $PROF $EXTRA > $R/gpurun_out/prof_$mode.log 2>&1)
    f=$(find gpurun_out/prof_$mode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r5_bench_cfg3_kernel_stats_$mode.csv
    python tools/trace_union.py gpurun_out/prof_$mode > gpurun_out/r5_trace_union_$mode.txt 2>&1
    rm -rf gpurun_out/prof_$mode
    grep "timed region" gpurun_out/prof_$mode.log | cut -c1-160 >> $S
  done
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_$c
    (cd /tmp && timeout 600 rocprofv3 --pmc $c -d $R/gpurun_out/pmc_$c -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-secondary --adapter-streams 0 --wgrad-stream 0 > $R/gpurun_out/pmc_$c.log 2>&1)
  done
  python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/r5_pmc_traffic.json >> $S 2>&1
  rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
  table encoder_gemms.py "encoder GEMMs"
  table decoder_gemms.py "decoder GEMMs, tiles 9 10" 9 10
  echo "== encoder step" >> $S
  timeout 300 python tools/encoder_step.py > gpurun_out/r5_encoder_step.json 2> gpurun_out/encoder_step.err
  cat gpurun_out/r5_encoder_step.json >> $S
  python tools/hbm_probe.py >> $S 2>&1
  ;;
epivalu)   # epilogue VALU diet (aux_grad fixed per tile, one med3 for the tail, fma column sums): tests, tables and the line, previous build beside it
  PREV=$R/multimae_amd/libmmae_hip_prev.so
  timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_h16_gpu.py -x -q -k "gelu or mlp or gemm or colsum or epilogue" 2>&1 | tail -3 >> $S
  MMAE_LIB=$PREV table encoder_gemms.py "encoder GEMMs, previous build"
  table encoder_gemms.py "encoder GEMMs, this build"
  MMAE_LIB=$PREV table decoder_gemms.py "decoder GEMMs, previous build" 9 10
  table decoder_gemms.py "decoder GEMMs, this build" 9 10
  MMAE_LIB=$PREV run "previous build" timeout 300 $B
  run "this build" timeout 300 $B
  MMAE_LIB=$PREV run "previous build again" timeout 300 $B
  run "this build again" timeout 300 $B
  ;;
dephase2)   # slack-aware start offsets (MMAE_PP_DEPHASE = n + 256 * mode, gemm_pp_body.h; experiments library: the production one reads no environment)
  [ -f $EXP ] || { echo "the experiments library did not travel: comment its line out of .gpurunignore (and run make -C multimae_amd/csrc exp)" >> $S; exit 1; }
  export MMAE_LIB=$EXP MMAE_TABLE_REP=15
  for cfg in 0 260 0 260 258 264 4 516; do
    MMAE_PP_DEPHASE=$cfg table encoder_gemms.py "encoder GEMMs, MMAE_PP_DEPHASE=$cfg"
  done
  ;;
pptrace)   # phase stamps inside the ping-pong GEMM (trace build: make -C multimae_amd/csrc trace)
  T=$R/multimae_amd/libmmae_hip_trace.so
  [ -f $T ] || { echo "no trace library: make -C multimae_amd/csrc trace" >> $S; exit 1; }
  MMAE_LIB=$T timeout 300 python tools/pp_trace.py >> $S 2>&1
  echo "#### the D = 256 block of an output adapter (50 176 rows)" >> $S
  MMAE_LIB=$T timeout 300 python tools/pp_trace.py --decoder >> $S 2>&1
  ;;
launch)   # the driver's launch forms on the one GPU a lease has: torchrun at world size 1, and two ranks sharing the device over gloo
  echo "== torchrun, world size 1 (the driver's N > 1 command line with N = 1)" >> $S
  ( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/launch1.log 2> gpurun_out/launch1.err ); echo "rc=$?" >> $S
  grep "timed region" gpurun_out/launch1.err | cut -c1-160 >> $S; grep "^{" gpurun_out/launch1.log | cut -c1-300 >> $S; grep -i "error\|Traceback" gpurun_out/launch1.err | tail -3 >> $S
  echo "== two ranks on one device, gloo buckets (functional test of the N > 1 path with the real kernels)" >> $S
  ( timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --share-device 1 --batch 64 > gpurun_out/launch2.log 2> gpurun_out/launch2.err ); echo "rc=$?" >> $S
  grep "timed region" gpurun_out/launch2.err | cut -c1-160 >> $S; grep "^{" gpurun_out/launch2.log | cut -c1-700 >> $S; grep -i "error\|Traceback" gpurun_out/launch2.err | tail -3 >> $S
  ;;
dropout)   # nn.Dropout sites: kernel, modules, the reference-recorded step; stochastic depth beside them
  timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_parity_geometry_gpu.py -x -q -k "dropout or drop_path" 2>&1 | tail -12 >> $S
  ;;
retest)   # re-run of the tests that failed in the last full run + the default line
  timeout 1500 python -m pytest tests/test_dropin_loop_gpu.py tests/test_model_gpu.py tests/test_parity_geometry_gpu.py -x -q -k "reference_loop_with or first_write or fused_optimizer" 2>&1 | tail -6 >> $S
  [ -f gpurun_out/dropin_fast_loop.json ] && cp gpurun_out/dropin_fast_loop.json gpurun_out/r5_dropin_fast_loop.json
  ;;
gelu)   # after a change of the GELU epilogue: kernel + parity tests, the encoder table, the default line twice
  timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_parity_geometry_gpu.py tests/test_curves_gpu.py -x -q -k "gelu or mlp or gemm or per_tensor or bench_batch or curve" 2>&1 | tail -5 >> $S
  table encoder_gemms.py "encoder GEMMs"
  run "production library, defaults" timeout 300 $B
  run "production library, defaults again" timeout 300 $B
  ;;
kernels)   # the kernel-level test file + the first-write / optimiser tests
  timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_h16_gpu.py tests/test_mxfp8_gpu.py -q 2>&1 | tail -6 >> $S
  ;;
lnfuse)   # decoder LayerNorms as GEMM side outputs: kernel + adapter tests, then the policy A/B inside one visit
  timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_parity_geometry_gpu.py -x -q -k "layernorm_side or side_outputs or (per_tensor and cfg3)" 2>&1 | tail -6 >> $S
  run "defaults (LayerNorms fused)" timeout 300 $B
  MMAE_LN_FUSE=0 run "MMAE_LN_FUSE=0 (own launches)" timeout 300 $B
  run "defaults again" timeout 300 $B
  MMAE_LN_FUSE=0 run "MMAE_LN_FUSE=0 again" timeout 300 $B
  kstats fused
  python - >> $S <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/r5_lnfuse_kernel_stats_fused.csv')))
print('== serialized ms/step', round(sum(float(r['TotalDurationNs']) for r in rows) / 8e6, 3), ' launches/step', sum(int(r['Calls']) for r in rows) / 8)
for r in rows:
    if 'ln_fwd' in r['Name'] or 'Lb1EEEv' in r['Name'] or ', true>(GemmArgs' in r['Name'] or 'cast_f32' in r['Name']:
        print(f"{float(r['TotalDurationNs']) / 8e6:7.3f} ms/step  {int(r['Calls']) / 8:6.1f} calls  {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:140]}")
PY
  ;;
baseline)
  run "production library, defaults" timeout 300 $B
  table encoder_gemms.py "encoder GEMMs"
  ;;
*) echo "unknown visit $V" >> $S ;;
esac
cat $S
