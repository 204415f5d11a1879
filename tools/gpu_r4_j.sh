#!/bin/bash
# round 4, visit j: decoder GEMM shapes per tile code; drop-in loop host profile; GradScaler device contract test
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
S=gpurun_out/r4j_summary.txt
: > $S
timeout 600 python -m pytest tests/test_reference_loop_gpu.py -q --tb=short -p no:cacheprovider > gpurun_out/r4j_pytest.log 2>&1
tail -15 gpurun_out/r4j_pytest.log | grep -E "passed|failed|Error|assert" >> $S
rm -rf gpurun_out/decg
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/decg -o p --output-format csv -- python $R/tools/decoder_gemms.py > $R/gpurun_out/decg.log 2>&1)
python tools/decoder_gemms.py --parse gpurun_out/decg > gpurun_out/r4j_decoder_gemms.txt 2>&1
rm -rf gpurun_out/decg
cat gpurun_out/r4j_decoder_gemms.txt >> $S
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5"
run() { label=$1; shift; ( "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-160)" >> $S; tail -3 gpurun_out/x.err | grep -i "error\|Traceback" >> $S; }
run "default" timeout 300 $B
run "dropin-ddp" timeout 300 $B --dropin-ddp 1
timeout 300 python -m cProfile -o gpurun_out/dropin.prof bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5 --dropin-ddp 1 > gpurun_out/x.log 2> gpurun_out/x.err
python - >> $S <<'PY'
import pstats
p = pstats.Stats('gpurun_out/dropin.prof')
import io, sys
s = io.StringIO(); pstats.Stats('gpurun_out/dropin.prof', stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[-9000:])
s = io.StringIO(); pstats.Stats('gpurun_out/dropin.prof', stream=s).sort_stats('tottime').print_stats(25); print(s.getvalue()[-5000:])
PY
rm -f gpurun_out/dropin.prof
cat $S
