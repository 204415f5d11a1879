#!/bin/bash
# Round-4 visit A: the new / touched kernels' tests, cfg5 + cfg3 per-tensor parity (the delta fix), bench A/B lines, serialized kernel
# stats, per-GEMM table of an encoder block, encoder step.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
S=gpurun_out/r4a_summary.txt
: > $S
echo "== targeted tests" >> $S
timeout 900 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -x > gpurun_out/r4a_pytest_kernels.log 2>&1
tail -3 gpurun_out/r4a_pytest_kernels.log >> $S
timeout 900 python -m pytest tests/test_parity_geometry_gpu.py -q --tb=short -p no:cacheprovider -k "cfg5 or bench_geometry" > gpurun_out/r4a_pytest_geom.log 2>&1
tail -3 gpurun_out/r4a_pytest_geom.log >> $S
head -6 gpurun_out/grad_parity_cfg5_bf16.txt >> $S
head -5 gpurun_out/grad_parity_cfg3_bf16.txt >> $S
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_reference_loop_gpu.py -q --tb=short -p no:cacheprovider -x > gpurun_out/r4a_pytest_model.log 2>&1
tail -3 gpurun_out/r4a_pytest_model.log >> $S
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5"
run() { label=$1; shift; ( "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-120)" >> $S; }
echo "== bench" >> $S
run "cfg3 default" timeout 300 $B
run "cfg3 default (again)" timeout 300 $B
run "cfg3 serialized" timeout 300 $B --adapter-streams 0 --wgrad-stream 0
run "cfg3 --force-dist 1 (reserve 16 CUs)" timeout 300 $B --force-dist 1
run "cfg3 --force-dist 1 --gemm-cu-reserve 0" timeout 300 $B --force-dist 1 --gemm-cu-reserve 0
run "cfg5 bf16" timeout 400 $B --config cfg5 --precision bf16 --steps 10 --warmup 3
echo "== kernel stats (serialized)" >> $S
rm -rf gpurun_out/prof_serialized
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_serialized -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary --adapter-streams 0 --wgrad-stream 0 > $R/gpurun_out/prof_serialized.log 2>&1)
f=$(find gpurun_out/prof_serialized -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4a_kernel_stats_serialized.csv
rm -rf gpurun_out/prof_serialized
python - >> $S <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/r4a_kernel_stats_serialized.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('serialized: kernel ms per step (8 profiled steps):', round(tot / 8 / 1e6, 3), ' launches per step:', sum(int(r['Calls']) for r in rows) / 8)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:45]:
    print(f"{float(r['TotalDurationNs']) / 8 / 1e6:8.3f} ms/step {int(r['Calls']) / 8:7.1f} calls/step  {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:100]}")
PY
echo "== per-GEMM table of an encoder block" >> $S
rm -rf gpurun_out/encg
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/encg -o p --output-format csv -- python $R/tools/encoder_gemms.py > $R/gpurun_out/encg.log 2>&1)
python tools/encoder_gemms.py --parse gpurun_out/encg > gpurun_out/r4a_encoder_gemms.txt 2>&1
rm -rf gpurun_out/encg
cat gpurun_out/r4a_encoder_gemms.txt >> $S
echo "== encoder step" >> $S
timeout 300 python tools/encoder_step.py > gpurun_out/r4a_encoder_step.json 2> gpurun_out/x.err
cat gpurun_out/r4a_encoder_step.json >> $S
cat $S
