#!/bin/bash
# Copy the judged summaries of a gpu_profile_round.sh run from gpurun_out/ (scratch) into profiles/ (tracked).
# usage: bash tools/collect_profiles.sh <tag>
tag=${1:?tag}
cp gpurun_out/bench_$tag.json profiles/${tag}_bench.json
grep -v amdgpu.ids gpurun_out/bench_$tag.err > profiles/${tag}_per_layer.txt
{ echo "# DN_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --profile-steps 0 --no-cpu-baseline --alt-steps 0 (14 steps with the default launch tape: 2 eager + 2 warm-up + 1 recorded inside the capture, 1 replay + 1 eager of the bit-for-bit check, then 2 warm-up + 5 timed replays; Disp_vgg_BN b32 128x416; single-stream so that a kernel's duration is its own)"; head -45 gpurun_out/prof_$tag/${tag}_kernel_stats.csv; } > profiles/${tag}_kernel_stats.csv
[ -f gpurun_out/pmc_${tag}_traffic.json ] && cp gpurun_out/pmc_${tag}_traffic.json profiles/${tag}_pmc_traffic.json
ls -la profiles | grep $tag
