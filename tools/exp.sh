export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out; : > gpurun_out/exp.txt
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "attention or golden or known" 2>&1 | tail -2 >> gpurun_out/exp.txt
rm -rf gpurun_out/attn_p
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/attn_p -o p --output-format csv -- python $R/tools/attn_probe.py > /dev/null 2>&1)
python tools/attn_probe.py --parse gpurun_out/attn_p >> gpurun_out/exp.txt
python bench.py --no-cpu-baseline --no-kernel-timing 2>&1 | grep "timed region" >> gpurun_out/exp.txt
cat gpurun_out/exp.txt
