mkdir -p gpurun_out; : > gpurun_out/exp.txt
run() { echo "== $*" >> gpurun_out/exp.txt; env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing $EXTRA 2>&1 | grep -E "timed region" >> gpurun_out/exp.txt; }
run A=1
run MMAE_SPLITK_WGS=128
run MMAE_SPLITK_WGS=64
EXTRA="--adapter-streams 0 --wgrad-stream 0" run A=serialized
EXTRA="--adapter-streams 1 --wgrad-stream 0" run A=adapter_streams_only
timeout 300 python -m pytest tests/test_model_gpu.py -q -k "graph" -p no:cacheprovider 2>&1 | tail -3 >> gpurun_out/exp.txt
cat gpurun_out/exp.txt
