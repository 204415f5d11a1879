#!/bin/bash
# Round 6: the collection script of this round's GPU visits (same helpers as tools/gpu_r5.sh).  Usage on the GPU box:
#   gpurun -- 'bash tools/gpu_r6.sh <visit> [...]'
# Everything lands under gpurun_out/r6_<visit>_*; what is worth keeping is copied to profiles/ by hand (profiles/INDEX.md).
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
V=${1:-baseline}; shift
S=gpurun_out/r6_${V}_summary.txt
: > $S
EXP=$R/multimae_amd/libmmae_hip_exp.so
PREV=$R/multimae_amd/libmmae_hip_prev.so
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 20 --warmup 5"
run() { label=$1; shift; ( "$@" > gpurun_out/x.log 2> gpurun_out/x.err ); echo "$label: $(grep 'timed region' gpurun_out/x.err | sed 's/.*done: //' | cut -c1-70)" >> $S; grep -i "error\|Traceback" gpurun_out/x.err | tail -3 >> $S; }
table() { tool=$1; label=$2; shift 2; rm -rf gpurun_out/tbl; (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/tbl -o p --output-format csv -- python $R/tools/$tool "$@" > $R/gpurun_out/tbl.log 2>&1); echo "== $label" >> $S; python tools/$tool --parse gpurun_out/tbl "$@" >> $S 2>&1; rm -rf gpurun_out/tbl; }
kstats() { tag=$1; shift; rm -rf gpurun_out/prof; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary --adapter-streams 0 --wgrad-stream 0 "$@" > $R/gpurun_out/prof.log 2>&1); f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r6_${V}_kernel_stats_$tag.csv; rm -rf gpurun_out/prof; }

case $V in
probe1)   # two half-batch encoder pipelines side by side vs one that owns the chip; this box's encoder table and default line
  run "production library, defaults" timeout 300 $B
  echo "== two_pipe_probe (forward + backward)" >> $S
  timeout 300 python tools/two_pipe_probe.py >> $S 2>&1
  echo "== two_pipe_probe --fwd-only" >> $S
  timeout 300 python tools/two_pipe_probe.py --fwd-only >> $S 2>&1
  table encoder_gemms.py "encoder GEMMs"
  ;;
*)
  echo "unknown visit $V" >> $S
  ;;
esac
cat $S
