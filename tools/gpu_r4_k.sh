#!/bin/bash
# round 4, visit k: decoder GEMM shapes with the workgroup de-phasing experiment (MMAE_PP_DEPHASE), GradScaler overflow test
# NOTE: the MMAE_* switches below are read only by the EXPERIMENTS library (common.h mmae_env_int); this script ran them against the production library, where they are no-ops -- its 'A/B' lines compare a build with itself.  The valid A/B of the same switches is tools/gpu_r4_p.sh (MMAE_LIB=.../libmmae_hip_exp.so).
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
S=gpurun_out/r4k_summary.txt
: > $S
timeout 300 python -m pytest tests/test_reference_loop_gpu.py -q --tb=short -p no:cacheprovider -k "overflow" > gpurun_out/r4k_pytest.log 2>&1
tail -15 gpurun_out/r4k_pytest.log | grep -E "passed|failed|Error|assert" >> $S
for dp in 0 1 2 4; do
rm -rf gpurun_out/decg
(cd /tmp && MMAE_PP_DEPHASE=$dp timeout 200 rocprofv3 --kernel-trace -d $R/gpurun_out/decg -o p --output-format csv -- python $R/tools/decoder_gemms.py 9 10 > $R/gpurun_out/decg.log 2>&1)
echo "== MMAE_PP_DEPHASE=$dp" >> $S
python tools/decoder_gemms.py --parse gpurun_out/decg 9 10 >> $S 2>&1
rm -rf gpurun_out/decg
done
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5"
run() { label=$1; shift; ( "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-60)" >> $S; tail -3 gpurun_out/x.err | grep -i "error\|Traceback" >> $S; }
run "default" timeout 300 $B
MMAE_PP_DEPHASE=2 run "dephase 2" timeout 300 $B
cat $S
