#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
S=gpurun_out/r4f_summary.txt
: > $S
timeout 600 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -x -k "loss or cross_entropy or patch_domain" > gpurun_out/r4f_pytest.log 2>&1
tail -3 gpurun_out/r4f_pytest.log >> $S
timeout 600 python -m pytest tests/test_model_gpu.py -q --tb=short -p no:cacheprovider -x > gpurun_out/r4f_pytest_m.log 2>&1
tail -3 gpurun_out/r4f_pytest_m.log >> $S
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5"
run() { label=$1; shift; ( "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-60)" >> $S; tail -2 gpurun_out/x.err | grep -i "error\|Traceback" >> $S; }
run "default" timeout 300 $B
run "default (again)" timeout 300 $B
R=$PWD
rm -rf gpurun_out/prof_serialized
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_serialized -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary --adapter-streams 0 --wgrad-stream 0 > $R/gpurun_out/prof_serialized.log 2>&1)
f=$(find gpurun_out/prof_serialized -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4f_kernel_stats_serialized.csv
rm -rf gpurun_out/prof_serialized
python - >> $S <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/r4f_kernel_stats_serialized.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('serialized: kernel ms per step (8 profiled steps):', round(tot / 8 / 1e6, 3), ' launches per step:', sum(int(r['Calls']) for r in rows) / 8)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    if 'loss' in r['Name'] or 'ce_' in r['Name'] or 'f32x3' in r['Name'] or 'attn' in r['Name']:
        print(f"{float(r['TotalDurationNs']) / 8 / 1e6:8.3f} ms/step {int(r['Calls']) / 8:7.1f} calls/step  {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:100]}")
PY
cat $S
