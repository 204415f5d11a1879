#!/bin/bash
# Round-4 visit C: fp16-operand (TF32-class) products for the fp32 adapter: kernel tests, parity at cfg3 / cfg5, bench A/B against x3,
# serialized kernel stats, copy-kernel census.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
S=gpurun_out/r4c_summary.txt
: > $S
echo "== tests" >> $S
timeout 600 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -x -k "f32f16 or f32x3 or patch_domain or cross_entropy or pixel_losses" > gpurun_out/r4c_pytest_k.log 2>&1
tail -3 gpurun_out/r4c_pytest_k.log >> $S
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_reference_loop_gpu.py -q --tb=short -p no:cacheprovider > gpurun_out/r4c_pytest_model.log 2>&1
tail -4 gpurun_out/r4c_pytest_model.log >> $S
timeout 900 python -m pytest tests/test_parity_geometry_gpu.py -q --tb=short -p no:cacheprovider -k "(cfg5 and bf16) or (bench_geometry and cfg3 and bf16)" > gpurun_out/r4c_pytest_geom.log 2>&1
tail -3 gpurun_out/r4c_pytest_geom.log >> $S
grep -n "semseg" gpurun_out/grad_parity_cfg3_bf16.txt | head -8 >> $S
head -4 gpurun_out/grad_parity_cfg5_bf16.txt >> $S
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5"
run() { label=$1; shift; ( "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-120)" >> $S; tail -2 gpurun_out/x.err | grep -i "error\|Traceback" >> $S; }
echo "== bench A/B" >> $S
run "cfg3 f16 adapter products (default)" timeout 300 $B
run "cfg3 x3 adapter products" timeout 300 $B --fp32-adapter-gemm x3
run "cfg3 f16 (again)" timeout 300 $B
run "cfg3 x3 (again)" timeout 300 $B --fp32-adapter-gemm x3
run "cfg3 f16 serialized" timeout 300 $B --adapter-streams 0 --wgrad-stream 0
echo "== kernel stats (serialized, f16)" >> $S
rm -rf gpurun_out/prof_serialized
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_serialized -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary --adapter-streams 0 --wgrad-stream 0 > $R/gpurun_out/prof_serialized.log 2>&1)
f=$(find gpurun_out/prof_serialized -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4c_kernel_stats_serialized.csv
python tools/copy_census.py gpurun_out/prof_serialized 8 > gpurun_out/r4c_copy_census.txt 2>&1
rm -rf gpurun_out/prof_serialized
python - >> $S <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/r4c_kernel_stats_serialized.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('serialized: kernel ms per step (8 profiled steps):', round(tot / 8 / 1e6, 3), ' launches per step:', sum(int(r['Calls']) for r in rows) / 8)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:24]:
    print(f"{float(r['TotalDurationNs']) / 8 / 1e6:8.3f} ms/step {int(r['Calls']) / 8:7.1f} calls/step  {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:100]}")
PY
head -30 gpurun_out/r4c_copy_census.txt >> $S
cat $S
