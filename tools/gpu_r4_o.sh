#!/bin/bash
# round 4, visit o: attention forward, grouped online-softmax form for > 128 keys at head_dim 32 (A/B via MMAE_ATTN_FWD_GRP)
# NOTE: the MMAE_* switches below are read only by the EXPERIMENTS library (common.h mmae_env_int); this script ran them against the production library, where they are no-ops -- its 'A/B' lines compare a build with itself.  The valid A/B of the same switches is tools/gpu_r4_p.sh (MMAE_LIB=.../libmmae_hip_exp.so).
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
S=gpurun_out/r4o_summary.txt
: > $S
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_h16_gpu.py tests/test_mxfp8_gpu.py -q --tb=short -p no:cacheprovider -k "attention or attn or adapter or block" > gpurun_out/r4o_pytest.log 2>&1
tail -12 gpurun_out/r4o_pytest.log | grep -E "passed|failed|FAILED|Error|assert" >> $S
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5"
run() { label=$1; shift; ( "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-60)" >> $S; tail -3 gpurun_out/x.err | grep -i "error\|Traceback" >> $S; }
run "GRP on" timeout 300 $B
export MMAE_ATTN_FWD_GRP=0; run "GRP off" timeout 300 $B; unset MMAE_ATTN_FWD_GRP
run "GRP on again" timeout 300 $B
for g in 1 0; do
rm -rf gpurun_out/prof_o
(cd /tmp && MMAE_ATTN_FWD_GRP=$g timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_o -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary --adapter-streams 0 --wgrad-stream 0 > $R/gpurun_out/prof_o.log 2>&1)
f=$(find gpurun_out/prof_o -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4o_kernel_stats_$g.csv
rm -rf gpurun_out/prof_o
python - $g >> $S <<'PY'
import csv, sys
rows = list(csv.DictReader(open(f'gpurun_out/r4o_kernel_stats_{sys.argv[1]}.csv')))
print('GRP', sys.argv[1], 'serialized ms/step', sum(float(r['TotalDurationNs']) for r in rows) / 8e6)
for r in rows:
    if 'attn_fwd' in r['Name']:
        print(r['Name'][:86], int(r['Calls']) / 8, round(float(r['AverageNs']) / 1e3, 1))
PY
done
cat $S
