#!/bin/bash
# fused MX quantisation: parity, then cfg5 step fused / unfused / bf16 on one box, MFMA PMC of the MX products
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mxfp8_gpu.py -x -q 2>&1 | grep -v "^    \|^$" | tail -12 > gpurun_out/mx_tests.log
cat gpurun_out/mx_tests.log
: > gpurun_out/mx_step.log
run() { # label, env..., args
  label="$1"; shift
  echo "== $label" >> gpurun_out/mx_step.log
  env "$@" > gpurun_out/x.log 2> gpurun_out/x.err
  grep "timed region" gpurun_out/x.err | tail -1 | cut -c1-200 >> gpurun_out/mx_step.log
  tail -1 gpurun_out/x.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print({k:r.get(k) for k in ('achieved','frac','gemm_ms_per_step','launches_per_step')}, r.get('mxfp8'))" >> gpurun_out/mx_step.log 2>&1
}
B="timeout 600 python bench.py --no-cpu-baseline --steps 15 --warmup 4"
run "cfg5 bf16" A=1 $B --config cfg5 --precision bf16
run "cfg5 mxfp8 (fused quantisation)" A=1 $B --config cfg5 --precision mxfp8
run "cfg5 mxfp8, MMAE_MX_FUSE=0 (separate passes)" MMAE_MX_FUSE=0 $B --config cfg5 --precision mxfp8
run "cfg3 mxfp8 (fused quantisation)" A=1 $B --config cfg3 --precision mxfp8
run "cfg3 bf16" A=1 $B --config cfg3 --precision bf16
cat gpurun_out/mx_step.log
timeout 300 python tools/mx_gemm_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/mx_gemm_bench.txt
rm -rf gpurun_out/pmc_mx
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_mx -o p --output-format csv -- python $R/tools/mx_gemm_bench.py > $R/gpurun_out/pmc_mx.log 2>&1)
python tools/pmc_mfma.py gpurun_out/pmc_mx > gpurun_out/pmc_mfma_mx.txt 2>&1
rm -rf gpurun_out/pmc_mx
cat gpurun_out/pmc_mfma_mx.txt
rm -rf gpurun_out/prof_cfg5_mxfp8
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cfg5_mxfp8 -o p --output-format csv -- python $R/bench.py --config cfg5 --precision mxfp8 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_cfg5_mxfp8.log 2>&1)
f=$(find gpurun_out/prof_cfg5_mxfp8 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_cfg5_mxfp8.csv
rm -rf gpurun_out/prof_cfg5_mxfp8
