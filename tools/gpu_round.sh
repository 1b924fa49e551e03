#!/bin/bash
# GPU-box helper: per-layer bench + rocprofv3 kernel stats.  usage: bash tools/gpu_round.sh <tag> [config]
tag=${1:-x}
cfg=${2:-vggbn128}
R=$(pwd)
mkdir -p gpurun_out
extra=""
if [ "$cfg" != "vggbn128" ]; then extra="--no-cpu-baseline --alt-steps 0"; fi
python bench.py --config $cfg --steps 10 --warmup 3 --per-layer $extra > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
cd /tmp; export TMPDIR=/tmp
# per-kernel durations are taken single-stream (DN_WGRAD_STREAM=0), like the instrumented steps bench.py's roofline comes from:
# with the weight gradients on the side stream two kernels share the chip and a kernel's wall time is not its own
DN_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o $tag --output-format csv -- python $R/bench.py --config $cfg --steps 5 --warmup 2 --profile-steps 0 --no-cpu-baseline --alt-steps 0 > $R/gpurun_out/prof_$tag.log 2>&1
cd $R
ls -R gpurun_out/prof_$tag > gpurun_out/prof_${tag}_summary.txt 2>&1 || true
