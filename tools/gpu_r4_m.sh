#!/bin/bash
# round 4, visit m: cross-entropy kernels (online softmax in groups of eight / four loads in flight), the stand-alone fp16-storage adapter test
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
S=gpurun_out/r4m_summary.txt
: > $S
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_h16_gpu.py tests/test_model_gpu.py -q --tb=short -p no:cacheprovider -k "loss or cross_entropy or patch_domain or standalone or known_answer or golden" > gpurun_out/r4m_pytest.log 2>&1
tail -12 gpurun_out/r4m_pytest.log | grep -E "passed|failed|FAILED|Error|assert" >> $S
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5"
run() { label=$1; shift; ( "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-60)" >> $S; tail -3 gpurun_out/x.err | grep -i "error\|Traceback" >> $S; }
run "default" timeout 300 $B
rm -rf gpurun_out/prof_m
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_m -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary --adapter-streams 0 --wgrad-stream 0 > $R/gpurun_out/prof_m.log 2>&1)
f=$(find gpurun_out/prof_m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4m_kernel_stats_serialized.csv
rm -rf gpurun_out/prof_m
python - >> $S <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/r4m_kernel_stats_serialized.csv')))
print('serialized ms/step', sum(float(r['TotalDurationNs']) for r in rows) / 8e6)
for r in rows:
    if 'ce_' in r['Name'] or 'loss' in r['Name'] or 'dwgroup_kernel<true, false, false' in r['Name']:
        print(r['Name'][:86], int(r['Calls']) / 8, round(float(r['AverageNs']) / 1e3, 1))
PY
cat $S
