#!/bin/bash
export TMPDIR=/tmp
R=$PWD
rm -rf gpurun_out/encg
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/encg -o p --output-format csv -- python $R/tools/encoder_gemms.py > $R/gpurun_out/encg.log 2>&1)
python tools/encoder_gemms.py --parse gpurun_out/encg > gpurun_out/encoder_gemms.txt 2>&1
rm -rf gpurun_out/encg
cat gpurun_out/encoder_gemms.txt
