#!/bin/bash
# Where does the host time go when two ranks share one GPU over gloo?  (functional mode only)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m torch.distributed.run --no-python --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29573 \
  python -m cProfile -o gpurun_out/share_prof.bin bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --share-device 1 --batch 64 --no-kernel-timing 2>&1 | grep "timed region" | cut -c1-300
python - <<'PY'
import pstats
p = pstats.Stats('gpurun_out/share_prof.bin')
p.sort_stats('tottime').print_stats(18)
PY
