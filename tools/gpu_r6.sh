#!/bin/bash
# Round 6: the collection script of this round's GPU visits (same helpers as tools/gpu_r5.sh).  Usage on the GPU box:
#   gpurun -- 'bash tools/gpu_r6.sh <visit> [...]'
# Everything lands under gpurun_out/r6_<visit>_*; what is worth keeping is copied to profiles/ by hand (profiles/INDEX.md).
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
V=${1:-baseline}; shift
S=gpurun_out/r6_${V}_summary.txt
: > $S
EXP=$R/multimae_amd/libmmae_hip_exp.so
PREV=$R/multimae_amd/libmmae_hip_prev.so
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5"
run() { label=$1; shift; ( "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-70)" >> $S; grep -i "error\|Traceback" gpurun_out/x.err | tail -3 >> $S; }
table() { tool=$1; label=$2; shift 2; rm -rf gpurun_out/tbl; (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/tbl -o p --output-format csv -- python $R/tools/$tool "$@" > $R/gpurun_out/tbl.log 2>&1); echo "== $label" >> $S; python tools/$tool --parse gpurun_out/tbl "$@" >> $S 2>&1; rm -rf gpurun_out/tbl; }
kstats() { tag=$1; shift; rm -rf gpurun_out/prof; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary --adapter-streams 0 --wgrad-stream 0 "$@" > $R/gpurun_out/prof.log 2>&1); f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r6_${V}_kernel_stats_$tag.csv; rm -rf gpurun_out/prof; }

case $V in
probe1)   # two half-batch encoder pipelines side by side vs one that owns the chip; this box's encoder table and default line
  run "production library, defaults" timeout 300 $B
  echo "== two_pipe_probe (forward + backward)" >> $S
  timeout 300 python tools/two_pipe_probe.py >> $S 2>&1
  echo "== two_pipe_probe --fwd-only" >> $S
  timeout 300 python tools/two_pipe_probe.py --fwd-only >> $S 2>&1
  table encoder_gemms.py "encoder GEMMs"
  ;;
probe2)   # the duo structure (two 4-wave workgroups per CU, tile 11) on TODAY's encoder epilogues; start offsets of half a tile period
  [ -f $EXP ] || { echo "the experiments library did not travel: comment its line out of .gpurunignore (and run make -C multimae_amd/csrc exp)" >> $S; exit 1; }
  export MMAE_LIB=$EXP
  table encoder_gemms.py "encoder GEMMs, exp library, planner's tiles"
  MMAE_GEMM_TILE=11 table encoder_gemms.py "encoder GEMMs, duo (tile 11) where it applies"
  MMAE_GEMM_TILE=11 MMAE_PP_DEPHASE=2 table encoder_gemms.py "encoder GEMMs, duo, second half of the grid ~8 us late"
  MMAE_GEMM_TILE=11 MMAE_PP_DEPHASE=4 table encoder_gemms.py "encoder GEMMs, duo, second half of the grid ~16 us late"
  MMAE_PP_DEPHASE=2 table encoder_gemms.py "encoder GEMMs, ping-pong, odd workgroups ~8 us late"
  MMAE_PP_DEPHASE=4 table encoder_gemms.py "encoder GEMMs, ping-pong, odd workgroups ~16 us late"
  MMAE_PP_DEPHASE=260 table encoder_gemms.py "encoder GEMMs, ping-pong, the workgroups with a tile less ~16 us late"
  table encoder_gemms.py "encoder GEMMs, exp library, planner's tiles, again"
  ;;
ab)   # same-visit A/B of the production library against the kept previous build (multimae_amd/libmmae_hip_prev.so); optional pytest -k filter in $1
  [ -f $PREV ] || { echo "no previous library: cp multimae_amd/libmmae_hip.so multimae_amd/libmmae_hip_prev.so before the change, and un-ignore it in .gpurunignore" >> $S; exit 1; }
  if [ -n "$1" ]; then echo "== pytest -m gpu -k '$1'" >> $S; timeout 1500 python -m pytest tests -m gpu -x -q -k "$1" 2>&1 | tail -6 >> $S; fi
  MMAE_LIB=$PREV table encoder_gemms.py "encoder GEMMs, previous build"
  table encoder_gemms.py "encoder GEMMs, this build"
  MMAE_LIB=$PREV table decoder_gemms.py "decoder GEMMs, previous build" 9 10
  table decoder_gemms.py "decoder GEMMs, this build" 9 10
  MMAE_LIB=$PREV run "previous build" timeout 300 $B
  run "this build" timeout 300 $B
  MMAE_LIB=$PREV run "previous build" timeout 300 $B
  run "this build" timeout 300 $B
  MMAE_LIB=$PREV table encoder_gemms.py "encoder GEMMs, previous build, again"
  table encoder_gemms.py "encoder GEMMs, this build, again"
  ;;
xattn)   # the fused q/kv-projection + cross-attention launch: kernel test, stand-alone A/B, adapter-level parity with the switch on, step A/B
  echo "== pytest fused cross-attention kernel + deterministic embedding gradient" >> $S
  timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "fused_cross_attention or semseg_class_embedding" 2>&1 | tail -4 >> $S
  echo "== tools/xattn_fused_ab.py" >> $S
  timeout 300 python tools/xattn_fused_ab.py >> $S 2>&1
  echo "== adapter / model parity tests with MMAE_XATTN_FUSE=1" >> $S
  MMAE_XATTN_FUSE=1 timeout 1500 python -m pytest tests/test_parity_geometry_gpu.py tests/test_model_gpu.py -x -q -k "per_tensor or mask_token or adapter or golden or mini" 2>&1 | tail -4 >> $S
  run "three launches (default)" timeout 300 $B
  MMAE_XATTN_FUSE=1 run "fused cross-attention" timeout 300 $B
  run "three launches (default), again" timeout 300 $B
  MMAE_XATTN_FUSE=1 run "fused cross-attention, again" timeout 300 $B
  ;;
ab2)   # light same-visit A/B against the kept previous build: optional pytest -k filter, then the default line twice each
  [ -f $PREV ] || { echo "no previous library (multimae_amd/libmmae_hip_prev.so; un-ignore it in .gpurunignore)" >> $S; exit 1; }
  if [ -n "$1" ]; then echo "== pytest -m gpu -k '$1'" >> $S; timeout 1500 python -m pytest tests -m gpu -x -q -k "$1" 2>&1 | tail -6 >> $S; fi
  MMAE_LIB=$PREV run "previous build" timeout 300 $B
  run "this build" timeout 300 $B
  MMAE_LIB=$PREV run "previous build" timeout 300 $B
  run "this build" timeout 300 $B
  MMAE_LIB=$PREV run "previous build" timeout 300 $B
  run "this build" timeout 300 $B
  if [ -n "$2" ]; then echo "== $2 (previous build)" >> $S; MMAE_LIB=$PREV timeout 300 python $2 >> $S 2>&1; echo "== $2 (this build)" >> $S; timeout 300 python $2 >> $S 2>&1; fi
  ;;
tests)   # the whole GPU suite + smoke
  echo "== pytest -m gpu" >> $S
  timeout 3000 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r6_pytest_gpu.log 2>&1
  tail -8 gpurun_out/r6_pytest_gpu.log >> $S
  echo "== smoke" >> $S
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-220 >> $S
  for f in curve_cfg3_bf16_h16.json adapter_modes_train_cfg3.json mxfp8_trains_cfg5.json dropin_fast_loop.json; do [ -f gpurun_out/$f ] && cp gpurun_out/$f gpurun_out/r6_$f; done
  ;;
final)   # the round's measurement visit: default line (+ secondaries), other configurations, kernel statistics, PMC traffic, tables, encoder step
  echo "== bench (default flags)" >> $S
  timeout 1200 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err
  grep "timed region\|cpu_baseline\|encoder step" gpurun_out/bench.err | cut -c1-200 >> $S
  grep "^{" gpurun_out/bench.log | tail -1 > gpurun_out/r6_bench_cfg3.json
  cat gpurun_out/r6_bench_cfg3.json >> $S
  echo "== other configurations" >> $S
  run "cfg3 default, again" timeout 300 $B
  MMAE_XATTN_FUSE=1 run "cfg3, the adapters' cross-attention as one fused launch (MMAE_XATTN_FUSE=1)" timeout 300 $B
  run "cfg3 default, a third time" timeout 300 $B
  run "cfg3, fp32 adapter as f32 tensors with fp16 operands (--fp32-adapter-gemm f16)" timeout 300 $B --fp32-adapter-gemm f16
  run "cfg3, fp32 adapter with split-bf16 operands (--fp32-adapter-gemm x3)" timeout 300 $B --fp32-adapter-gemm x3
  run "cfg3 serialized (one stream)" timeout 300 $B --adapter-streams 0 --wgrad-stream 0
  run "cfg2 (RGB only) B=256" timeout 300 $B --config cfg2
  run "cfg5 geometry (ViT-L, bf16) B=128" timeout 600 $B --config cfg5 --precision bf16 --steps 10 --warmup 3
  timeout 600 python bench.py --no-cpu-baseline --no-secondary --config cfg5 --precision mxfp8 --steps 10 --warmup 3 > gpurun_out/bench_cfg5_mxfp8.log 2> gpurun_out/x.err
  grep "^{" gpurun_out/bench_cfg5_mxfp8.log | tail -1 > gpurun_out/r6_bench_cfg5_mxfp8.json
  echo "cfg5 (ViT-L, mxfp8) B=128: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-120)" >> $S
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5 --force-dist 1 > gpurun_out/dist_world1.log 2> gpurun_out/x.err
  echo "cfg3, data-parallel path over RCCL at world size 1, all_reduce buckets issued (--force-dist 1): $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-120)" >> $S
  grep "^{" gpurun_out/dist_world1.log | tail -1 > gpurun_out/r6_dist_world1_rccl.json
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5 --force-dist 1 --dp-exchange rs_ag > gpurun_out/dist_world1_rsag.log 2> gpurun_out/x.err
  echo "cfg3, the same with reduce_scatter + all_gather buckets (--dp-exchange rs_ag): $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-120)" >> $S
  grep -i "error\|Traceback" gpurun_out/x.err | tail -3 >> $S
  PROF="--steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary"
  for mode in default serialized; do
    EXTRA=""; [ $mode = serialized ] && EXTRA="--adapter-streams 0 --wgrad-stream 0"
    rm -rf gpurun_out/prof_$mode
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$mode -o p --output-format csv -- python $R/bench.py $PROF $EXTRA > $R/gpurun_out/prof_$mode.log 2>&1)
    f=$(find gpurun_out/prof_$mode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r6_bench_cfg3_kernel_stats_$mode.csv
    python tools/trace_union.py gpurun_out/prof_$mode > gpurun_out/r6_trace_union_$mode.txt 2>&1
    rm -rf gpurun_out/prof_$mode
    grep "timed region" gpurun_out/prof_$mode.log | cut -c1-160 >> $S
  done
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_$c
    (cd /tmp && timeout 600 rocprofv3 --pmc $c -d $R/gpurun_out/pmc_$c -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-secondary --adapter-streams 0 --wgrad-stream 0 > $R/gpurun_out/pmc_$c.log 2>&1)
  done
  python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/r6_pmc_traffic.json >> $S 2>&1
  rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
  table encoder_gemms.py "encoder GEMMs"
  table decoder_gemms.py "decoder GEMMs, tiles 9 10" 9 10
  echo "== encoder step" >> $S
  timeout 300 python tools/encoder_step.py > gpurun_out/r6_encoder_step.json 2> gpurun_out/encoder_step.err
  cat gpurun_out/r6_encoder_step.json >> $S
  python tools/hbm_probe.py >> $S 2>&1
  ;;
envab)   # same-visit A/B of an environment switch of the production path: $1 = VAR=value of the alternative, $2 = label
  for i in 1 2 3; do
    run "defaults" timeout 300 $B
    env $1 bash -c "$(declare -f run); S=$S; run \"$2 ($1)\" timeout 300 $B"
  done
  ;;
launch)   # the driver's launch forms on the one GPU a lease has: torchrun at world size 1, and two ranks sharing the device over gloo (all_reduce and rs_ag buckets)
  echo "== torchrun, world size 1 (the driver's N > 1 command line with N = 1)" >> $S
  ( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/launch1.log 2> gpurun_out/launch1.err ); echo "rc=$?" >> $S
  grep "timed region" gpurun_out/launch1.err | cut -c1-160 >> $S; grep "^{" gpurun_out/launch1.log | cut -c1-300 >> $S; grep -i "error\|Traceback" gpurun_out/launch1.err | tail -3 >> $S
  for ex in all_reduce rs_ag; do
    echo "== two ranks on one device, gloo buckets, --dp-exchange $ex (functional test of the N > 1 path with the real kernels)" >> $S
    ( timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --share-device 1 --batch 64 --dp-exchange $ex > gpurun_out/launch2.log 2> gpurun_out/launch2.err ); echo "rc=$?" >> $S
    grep "timed region" gpurun_out/launch2.err | cut -c1-160 >> $S; grep "^{" gpurun_out/launch2.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print({k: d.get(k) for k in ('value','n_gpus','ms_per_step','rccl_ranks_seen')}, {k: d['data_parallel'].get(k) for k in ('exchange','buckets','backend','exposed_allreduce_ms_per_step')})" >> $S 2>&1; grep -i "error\|Traceback" gpurun_out/launch2.err | tail -3 >> $S
  done
  ;;
width)   # how wide must the persistent GEMM grids be?  the default line with 0 / 16 / 32 / 64 CUs left free for the WHOLE step (the loops are power-bound)
  for k in 0 16 32 0 16 64 96; do run "GEMM grids leave $k CUs free" timeout 300 $B --gemm-cu-reserve $k; done
  ;;
copies)   # where do the __amd_rocclr_copyBuffer / fill launches of a step come from?  (tools/copy_census.py on a kernel trace of the bench command)
  rm -rf gpurun_out/prof_c
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_c -o p --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-secondary > $R/gpurun_out/prof_c.log 2>&1)
  python tools/copy_census.py gpurun_out/prof_c 6 >> $S 2>&1
  rm -rf gpurun_out/prof_c
  ;;
pptrace)   # phase stamps inside the ping-pong GEMM on this round's tile boundary (trace build: make -C multimae_amd/csrc trace; un-ignore libmmae_hip_trace.so)
  T=$R/multimae_amd/libmmae_hip_trace.so
  [ -f $T ] || { echo "no trace library: make -C multimae_amd/csrc trace" >> $S; exit 1; }
  MMAE_LIB=$T timeout 300 python tools/pp_trace.py >> $S 2>&1
  ;;
*)
  echo "unknown visit $V" >> $S
  ;;
esac
cat $S
