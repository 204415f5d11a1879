#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
: > gpurun_out/summary.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4 >> gpurun_out/summary.txt
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5"
run() { label=$1; shift; ( env "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-60)" >> gpurun_out/summary.txt; }
for v in "CUR=1" "MMAE_LIB=$R/multimae_amd/libmmae_hip_unr.so" "CUR=1" "MMAE_LIB=$R/multimae_amd/libmmae_hip_unr.so"; do
  run "$v default-streams" $v timeout 300 $B
  run "$v serialized" $v timeout 300 $B --adapter-streams 0 --wgrad-stream 0
done
for v in "CUR=1" "MMAE_LIB=$R/multimae_amd/libmmae_hip_unr.so"; do
  rm -rf gpurun_out/encg
  (cd /tmp && env $v timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/encg -o p --output-format csv -- python $R/tools/encoder_gemms.py > $R/gpurun_out/encg.log 2>&1)
  echo "== encoder_gemms $v" >> gpurun_out/summary.txt
  python tools/encoder_gemms.py --parse gpurun_out/encg >> gpurun_out/summary.txt 2>&1
done
rm -rf gpurun_out/encg
cat gpurun_out/summary.txt
