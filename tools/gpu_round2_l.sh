#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
: > gpurun_out/summary.txt
for v in "MMAE_PP_DEPHASE=0" "MMAE_PP_DEPHASE=1" "MMAE_PP_DEPHASE=2" "MMAE_PP_DEPHASE=3" "MMAE_PP_DEPHASE=5"; do
  rm -rf gpurun_out/encg
  (cd /tmp && env $v timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/encg -o p --output-format csv -- python $R/tools/encoder_gemms.py > $R/gpurun_out/encg.log 2>&1)
  echo "== encoder_gemms $v" >> gpurun_out/summary.txt
  python tools/encoder_gemms.py --parse gpurun_out/encg | head -8 >> gpurun_out/summary.txt 2>&1
done
rm -rf gpurun_out/encg
cat gpurun_out/summary.txt
