#!/bin/bash
# round 4, visit n: grouped weight-gradient reduction (problem-uniform workgroups, 4 float4 per thread), decoder_build (4 rows per workgroup)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
S=gpurun_out/r4n_summary.txt
: > $S
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_h16_gpu.py tests/test_model_gpu.py -q --tb=short -p no:cacheprovider -k "dw_group or decoder_build or adapter or block or stack or known_answer or golden or queries" > gpurun_out/r4n_pytest.log 2>&1
tail -12 gpurun_out/r4n_pytest.log | grep -E "passed|failed|FAILED|Error|assert" >> $S
timeout 600 python -m pytest tests/test_parity_geometry_gpu.py -q --tb=short -p no:cacheprovider -k "bench_geometry or queries or chunks" > gpurun_out/r4n_pytest_p.log 2>&1
tail -12 gpurun_out/r4n_pytest_p.log | grep -E "passed|failed|FAILED|Error|assert" >> $S
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5"
run() { label=$1; shift; ( "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-60)" >> $S; tail -3 gpurun_out/x.err | grep -i "error\|Traceback" >> $S; }
run "default" timeout 300 $B
run "default again" timeout 300 $B
rm -rf gpurun_out/prof_n
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_n -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary --adapter-streams 0 --wgrad-stream 0 > $R/gpurun_out/prof_n.log 2>&1)
f=$(find gpurun_out/prof_n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4n_kernel_stats_serialized.csv
rm -rf gpurun_out/prof_n
python - >> $S <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/r4n_kernel_stats_serialized.csv')))
print('serialized ms/step', sum(float(r['TotalDurationNs']) for r in rows) / 8e6)
for r in rows:
    if 'dw_group' in r['Name'] or 'decoder_build' in r['Name'] or 'dwgroup_kernel<true, false, false' in r['Name']:
        print(r['Name'][:86], int(r['Calls']) / 8, round(float(r['AverageNs']) / 1e3, 1))
PY
cat $S
