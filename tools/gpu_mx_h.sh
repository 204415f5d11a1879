#!/bin/bash
# HBM traffic (PMC) of a cfg5 step in mxfp8 mode, single stream
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcx_$c
  (cd /tmp && timeout 600 rocprofv3 --pmc $c -d $R/gpurun_out/pmcx_$c -o p --output-format csv -- python $R/bench.py --config cfg5 --precision mxfp8 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --adapter-streams 0 --wgrad-stream 0 > $R/gpurun_out/pmcx_$c.log 2>&1)
done
python tools/pmc_traffic.py gpurun_out/pmcx_FETCH_SIZE gpurun_out/pmcx_WRITE_SIZE gpurun_out/pmc_traffic_cfg5_mxfp8.json 2>&1 | cut -c1-200
rm -rf gpurun_out/pmcx_FETCH_SIZE gpurun_out/pmcx_WRITE_SIZE
