#!/bin/bash
# FETCH_SIZE of the x3 kernels in the bench step (one --pmc pass, serialized)
export TMPDIR=/tmp
R=$PWD
rm -rf gpurun_out/pmc_x3
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_x3 -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-secondary --adapter-streams 0 --wgrad-stream 0 > $R/gpurun_out/pmc_x3.log 2>&1)
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_x3/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: [0, 0.0])
for fn in f:
    for r in csv.DictReader(open(fn)):
        if 'gemm_f32x3' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
            a = acc[r['Kernel_Name'][:70]]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, (n, v) in acc.items():
    print(k, n, 'launches', round(v / n * 1024 * 2 / 1e6, 1), 'MB fetched per launch (KB units x2 gfx950 correction)')
PY
rm -rf gpurun_out/pmc_x3
