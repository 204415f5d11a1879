#!/bin/bash
# round 4, visit i: attention backward with the register-resident pass 1 (A/B via MMAE_ATTN_BWD_RS), cfg5 f16 vs h16, drop-in host time
# NOTE: the MMAE_* switches below are read only by the EXPERIMENTS library (common.h mmae_env_int); this script ran them against the production library, where they are no-ops -- its 'A/B' lines compare a build with itself.  The valid A/B of the same switches is tools/gpu_r4_p.sh (MMAE_LIB=.../libmmae_hip_exp.so).
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
S=gpurun_out/r4i_summary.txt
: > $S
timeout 900 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -k "attention or attn" > gpurun_out/r4i_pytest.log 2>&1
tail -5 gpurun_out/r4i_pytest.log | grep -E "passed|failed" >> $S
timeout 600 python -m pytest tests/test_mxfp8_gpu.py -q --tb=short -p no:cacheprovider -k "attn or attention or block or stack" > gpurun_out/r4i_pytest_mx.log 2>&1
tail -5 gpurun_out/r4i_pytest_mx.log | grep -E "passed|failed" >> $S
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5"
run() { label=$1; shift; ( "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-160)" >> $S; tail -3 gpurun_out/x.err | grep -i "error\|Traceback" >> $S; }
run "default (RS on)" timeout 300 $B
MMAE_ATTN_BWD_RS=0 run "RS off" timeout 300 $B
run "default (RS on) again" timeout 300 $B
export MMAE_ATTN_BWD_RS=0; run "RS off again" timeout 300 $B; unset MMAE_ATTN_BWD_RS
run "cfg5 bf16 h16" timeout 600 $B --config cfg5 --precision bf16 --steps 10 --warmup 3
run "cfg5 bf16 f16" timeout 600 $B --config cfg5 --precision bf16 --steps 10 --warmup 3 --fp32-adapter-gemm f16
run "dropin-ddp" timeout 300 $B --dropin-ddp 1
rm -rf gpurun_out/prof_i
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_i -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary --adapter-streams 0 --wgrad-stream 0 > $R/gpurun_out/prof_i.log 2>&1)
f=$(find gpurun_out/prof_i -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4i_kernel_stats_serialized.csv
rm -rf gpurun_out/prof_i
grep attn gpurun_out/r4i_kernel_stats_serialized.csv | awk -F, '{print $1, $2, $4}' | cut -c1-150 >> $S
cat $S
