#!/bin/bash
# rocprofv3 kernel stats of the production step, serialized (single stream) and default (multi-stream): gpurun_out/kernel_stats_{serialized,default}.csv
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
for mode in serialized default; do
  if [ $mode = serialized ]; then FLAGS="--adapter-streams 0 --wgrad-stream 0"; else FLAGS=""; fi
  rm -rf gpurun_out/prof_$mode
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$mode -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing $FLAGS > $R/gpurun_out/prof_$mode.log 2>&1)
  f=$(find gpurun_out/prof_$mode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_$mode.csv
  rm -rf gpurun_out/prof_$mode
  grep "timed region" gpurun_out/prof_$mode.log
done
python - <<'PY'
import csv
for mode in ('serialized',):
    rows = list(csv.DictReader(open(f'gpurun_out/kernel_stats_{mode}.csv')))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print(mode, 'total kernel ms per step (8 profiled steps incl. warmup):', tot / 8 / 1e6, ' launches per step:', sum(int(r['Calls']) for r in rows) / 8)
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:32]:
        print(f"{float(r['TotalDurationNs']) / 8 / 1e6:8.3f} ms/step {int(r['Calls']) / 8:7.1f} calls/step  {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:110]}")
PY
