#!/bin/bash
# Round-6 profile set in one GPU call: the whole GPU suite, the driver's default bench call (headline + other_configs + strong_1gpu),
# rocprofv3 kernel stats + per-layer table + PMC traffic of the headline config, fresh SQ counters of the dominant kernel, the 2-rank
# gloo plumbing of bench.py on the one GPU (own-communicator default must fall back to torch.distributed there).
# usage (GPU box): bash tools/gpu_profile_round6.sh <tag> [notests]
tag=${1:-r06_b}
mkdir -p gpurun_out
exec < /dev/null
if [ "$2" != "notests" ]; then
  python -m pytest tests -q -m gpu > gpurun_out/tests_$tag.log 2>&1; grep -E "passed|failed|error" gpurun_out/tests_$tag.log | tail -3
fi
python bench.py > gpurun_out/bench_default_$tag.json 2> gpurun_out/bench_default_$tag.err
bash tools/pmc_traffic.sh $tag
bash tools/gpu_round.sh $tag
bash tools/pmc_micro.sh ${tag}_wino8 c512_512_16x52,c256_256_32x104 fwd,dgrad > gpurun_out/sq_${tag}_wino8.txt 2>&1
DN_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --alt-steps 0 --profile-steps 0 > gpurun_out/bench_${tag}_2ranks_gloo.json 2> gpurun_out/bench_${tag}_2ranks_gloo.err
tail -2 gpurun_out/bench_${tag}_2ranks_gloo.err
