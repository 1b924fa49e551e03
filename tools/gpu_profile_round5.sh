#!/bin/bash
# Round-5 profile set in one GPU call: the whole GPU test suite, the bench line (+ per layer), rocprofv3 kernel stats + PMC traffic of the
# headline config, kernel stats of photo128 / dorn128, the other configs, the one-GPU strong-scaling table incl. the data-parallel
# machinery on a single-rank communicator, a 4-image timeline, fresh SQ counters of the two Winograd forward kernels.
# usage (GPU box): bash tools/gpu_profile_round5.sh <tag> [notests]
tag=${1:-r05_a}
mkdir -p gpurun_out
exec < /dev/null
if [ "$2" != "notests" ]; then
  python -m pytest tests -q -m gpu > gpurun_out/tests_$tag.log 2>&1; grep -E "passed|failed|error" gpurun_out/tests_$tag.log | tail -3
fi
python bench.py > gpurun_out/bench_default_$tag.json 2> gpurun_out/bench_default_$tag.err
bash tools/pmc_traffic.sh $tag
bash tools/gpu_round.sh $tag
bash tools/gpu_round.sh ${tag}_photo128 photo128
bash tools/gpu_round.sh ${tag}_dorn128 dorn128
python bench.py --config res50_480 --steps 10 --warmup 3 --per-layer --no-cpu-baseline --alt-steps 0 > gpurun_out/bench_${tag}_res50_480.json 2> gpurun_out/bench_${tag}_res50_480_per_layer.txt
python bench.py --config vggbn480 --steps 10 --warmup 3 --no-cpu-baseline --alt-steps 0 > gpurun_out/bench_${tag}_vggbn480.json 2>/dev/null
python bench.py --config dorn128 --compute bf16 --steps 10 --warmup 3 --no-cpu-baseline --alt-steps 0 > gpurun_out/bench_${tag}_dorn128_bf16.json 2>/dev/null
bash tools/strong_scaling_1gpu.sh gpurun_out/strong_$tag "tape eager rccl1 rccl1+standin" > /dev/null 2>&1
cp gpurun_out/strong_$tag/summary.txt gpurun_out/strong_${tag}.txt
bash tools/step_timeline.sh 4 gpurun_out/timeline_$tag > /dev/null 2>&1
bash tools/pmc_micro.sh ${tag}_wino8 c512_512_16x52,c256_256_32x104 fwd,dgrad > gpurun_out/sq_${tag}_wino8.txt 2>&1
bash tools/pmc_micro.sh ${tag}_wino4 c64_64_128x416 fwd,dgrad > gpurun_out/sq_${tag}_wino64.txt 2>&1
tail -3 gpurun_out/strong_${tag}.txt
