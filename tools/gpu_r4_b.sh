#!/bin/bash
# Round-4 visit B: two-launch batched column sums, lazy predictions, DDP drop-in path; bench A/B; PMC dissection (experiments library);
# MX-fp8 training curve.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
S=gpurun_out/r4b_summary.txt
: > $S
echo "== tests" >> $S
timeout 600 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -x -k "colsum or block_backward or epilogue or gelu" > gpurun_out/r4b_pytest_k.log 2>&1
tail -3 gpurun_out/r4b_pytest_k.log >> $S
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_reference_loop_gpu.py -q --tb=short -p no:cacheprovider > gpurun_out/r4b_pytest_model.log 2>&1
tail -5 gpurun_out/r4b_pytest_model.log >> $S
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5"
run() { label=$1; shift; ( "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-120)" >> $S; tail -2 gpurun_out/x.err | grep -i "error\|Traceback" >> $S; }
echo "== bench" >> $S
run "cfg3 default" timeout 300 $B
run "cfg3 default (again)" timeout 300 $B
run "cfg3 serialized" timeout 300 $B --adapter-streams 0 --wgrad-stream 0
run "cfg3 drop-in DDP loop (--dropin-ddp 1)" timeout 300 $B --dropin-ddp 1
run "cfg3 --force-dist 1 (reserve 16 CUs)" timeout 300 $B --force-dist 1
run "cfg3 --force-dist 1 --gemm-cu-reserve 0" timeout 300 $B --force-dist 1 --gemm-cu-reserve 0
echo "== encoder step" >> $S
timeout 300 python tools/encoder_step.py > gpurun_out/r4b_encoder_step.json 2> gpurun_out/x.err
cat gpurun_out/r4b_encoder_step.json >> $S
echo "== kernel stats (serialized)" >> $S
rm -rf gpurun_out/prof_serialized
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_serialized -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary --adapter-streams 0 --wgrad-stream 0 > $R/gpurun_out/prof_serialized.log 2>&1)
f=$(find gpurun_out/prof_serialized -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4b_kernel_stats_serialized.csv
rm -rf gpurun_out/prof_serialized
python - >> $S <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/r4b_kernel_stats_serialized.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('serialized: kernel ms per step (8 profiled steps):', round(tot / 8 / 1e6, 3), ' launches per step:', sum(int(r['Calls']) for r in rows) / 8)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:30]:
    print(f"{float(r['TotalDurationNs']) / 8 / 1e6:8.3f} ms/step {int(r['Calls']) / 8:7.1f} calls/step  {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:100]}")
PY
echo "== PMC dissection (experiments library)" >> $S
rm -rf gpurun_out/pmcd; mkdir -p gpurun_out/pmcd
(cd /tmp && MMAE_LIB=$R/multimae_amd/libmmae_hip_exp.so MMAE_EXPERIMENTS=1 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmcd -o p --output-format csv -- python $R/tools/pmc_dissection.py --plan $R/gpurun_out/pmcd/plan.json > $R/gpurun_out/pmcd.log 2>&1)
python tools/pmc_dissection.py --parse gpurun_out/pmcd > gpurun_out/r4b_pmc_dissection.txt 2>&1
cat gpurun_out/r4b_pmc_dissection.txt >> $S
tail -3 gpurun_out/pmcd.log >> $S
rm -rf gpurun_out/pmcd
echo "== MX-fp8 trains?" >> $S
timeout 900 python tools/mx_trains.py > gpurun_out/mx_trains.log 2>&1
tail -4 gpurun_out/mx_trains.log >> $S
cat $S
