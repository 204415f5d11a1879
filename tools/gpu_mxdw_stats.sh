#!/bin/bash
# cfg5 / mxfp8 step, weight gradients bf16 (MMAE_MX_WGRAD=0) against MX-fp8 (1): serialized kernel stats of the kernels involved
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
for w in 0 1; do
  rm -rf gpurun_out/prof_mxdw$w
  (cd /tmp && MMAE_MX_WGRAD=$w timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_mxdw$w -o p --output-format csv -- python $R/bench.py --config cfg5 --precision mxfp8 --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary --adapter-streams 0 --wgrad-stream 0 > $R/gpurun_out/prof_mxdw$w.log 2>&1)
  f=$(find gpurun_out/prof_mxdw$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_cfg5_mxdw$w.csv
  rm -rf gpurun_out/prof_mxdw$w
  grep "timed region" gpurun_out/prof_mxdw$w.log | cut -c1-100
done
python - <<'PY'
import csv
for w in (0, 1):
    rows = list(csv.DictReader(open(f'gpurun_out/kernel_stats_cfg5_mxdw{w}.csv')))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print(f'== MMAE_MX_WGRAD={w}: total kernel ms per step', round(tot / 8 / 1e6, 2), ' launches per step', sum(int(r['Calls']) for r in rows) / 8)
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:14]:
        print(f"{float(r['TotalDurationNs']) / 8 / 1e6:8.3f} ms/step {int(r['Calls']) / 8:7.1f} calls/step  {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:100]}")
PY
