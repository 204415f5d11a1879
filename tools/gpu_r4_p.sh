#!/bin/bash
# round 4, visit p: the A/B switches live only in the EXPERIMENTS build (common.h: mmae_env_int) -- attention backward RS, attention forward GRP and the
# ping-pong workgroup de-phasing, each against its default, all through multimae_amd/libmmae_hip_exp.so (make -C multimae_amd/csrc exp)
mkdir -p gpurun_out
export TMPDIR=/tmp
export MMAE_LIB=$PWD/multimae_amd/libmmae_hip_exp.so
R=$PWD
S=gpurun_out/r4p_summary.txt
: > $S
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5"
run() { label=$1; shift; ( "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-60)" >> $S; tail -3 gpurun_out/x.err | grep -i "error\|Traceback" >> $S; }
run "defaults (exp library)" timeout 300 $B
MMAE_ATTN_BWD_RS=0 run "MMAE_ATTN_BWD_RS=0" timeout 300 $B
MMAE_ATTN_FWD_GRP=0 run "MMAE_ATTN_FWD_GRP=0" timeout 300 $B
run "defaults again" timeout 300 $B
MMAE_ATTN_BWD_RS=0 MMAE_ATTN_FWD_GRP=0 run "both off" timeout 300 $B
MMAE_PP_DEPHASE=2 run "MMAE_PP_DEPHASE=2" timeout 300 $B
for cfg in "A " "B MMAE_ATTN_BWD_RS=0 MMAE_ATTN_FWD_GRP=0"; do
tag=${cfg%% *}; envs=${cfg#* }
rm -rf gpurun_out/prof_p
(cd /tmp && env $envs timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_p -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary --adapter-streams 0 --wgrad-stream 0 > $R/gpurun_out/prof_p.log 2>&1)
f=$(find gpurun_out/prof_p -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4p_kernel_stats_$tag.csv
rm -rf gpurun_out/prof_p
python - $tag "$envs" >> $S <<'PY'
import csv, sys
rows = list(csv.DictReader(open(f'gpurun_out/r4p_kernel_stats_{sys.argv[1]}.csv')))
print('==', sys.argv[2] or 'defaults', ': serialized ms/step', round(sum(float(r['TotalDurationNs']) for r in rows) / 8e6, 3))
for r in rows:
    if 'attn_' in r['Name']:
        print('  ', r['Name'][28:86], int(r['Calls']) / 8, round(float(r['AverageNs']) / 1e3, 1))
PY
done
for dp in 0 2; do
rm -rf gpurun_out/decg
(cd /tmp && MMAE_PP_DEPHASE=$dp timeout 200 rocprofv3 --kernel-trace -d $R/gpurun_out/decg -o p --output-format csv -- python $R/tools/decoder_gemms.py 9 10 > $R/gpurun_out/decg.log 2>&1)
echo "== decoder GEMMs, MMAE_PP_DEPHASE=$dp" >> $S
python tools/decoder_gemms.py --parse gpurun_out/decg 9 10 >> $S 2>&1
rm -rf gpurun_out/decg
done
cat $S
