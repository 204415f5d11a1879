#!/bin/bash
# GPU visit C: grouped weight gradients (tests + bench A/B), epilogue dissection of the GELU product.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
: > gpurun_out/summary.txt
echo "== pytest (kernels + parity geometry)" >> gpurun_out/summary.txt
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_parity_geometry_gpu.py tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
tail -15 gpurun_out/pytest_gpu.log >> gpurun_out/summary.txt
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5"
run() { label=$1; shift; ( env "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //')" >> gpurun_out/summary.txt; }
run "dw_group=1 (default)" timeout 300 $B
run "dw_group=0" MMAE_DW_GROUP=0 timeout 300 $B
run "dw_group=1 serialized" timeout 300 $B --adapter-streams 0 --wgrad-stream 0
run "dw_group=0 serialized" MMAE_DW_GROUP=0 timeout 300 $B --adapter-streams 0 --wgrad-stream 0
run "dw_group=1 cfg2" timeout 300 $B --config cfg2
echo "== encoder step" >> gpurun_out/summary.txt
timeout 300 python tools/encoder_step.py >> gpurun_out/summary.txt 2> gpurun_out/encoder_step.err
MMAE_DW_GROUP=0 timeout 300 python tools/encoder_step.py >> gpurun_out/summary.txt 2>> gpurun_out/encoder_step.err
for v in "MMAE_EPI_DBG=0" "MMAE_EPI_DBG=1" "MMAE_EPI_DBG=2" "MMAE_EPI_DBG=3" "MMAE_EPI_DBG=4" "MMAE_PP_PERSIST=0" "MMAE_GEMM_XCD=0"; do
  rm -rf gpurun_out/encg
  (cd /tmp && env $v timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/encg -o p --output-format csv -- python $R/tools/encoder_gemms.py > $R/gpurun_out/encg.log 2>&1)
  echo "== encoder_gemms $v" >> gpurun_out/summary.txt
  python tools/encoder_gemms.py --parse gpurun_out/encg >> gpurun_out/summary.txt 2>&1
done
rm -rf gpurun_out/encg
cat gpurun_out/summary.txt
