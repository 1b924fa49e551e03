#!/bin/bash
# Round-4 profile set in one GPU call: tests, bench line (+ per layer), rocprofv3 kernel stats + PMC traffic for the headline config AND for
# photo128 / dorn128 (the loss / warp / ordinal kernels north_star names), other configs, one-GPU strong scaling.
# usage (GPU box): bash tools/gpu_profile_round4.sh <tag> [notests]
tag=${1:-r04_a}
mkdir -p gpurun_out
if [ "$2" != "notests" ]; then
  python -m pytest tests -q -m gpu > gpurun_out/tests_$tag.log 2>&1; tail -3 gpurun_out/tests_$tag.log
fi
bash tools/pmc_traffic.sh $tag
bash tools/gpu_round.sh $tag
bash tools/pmc_traffic.sh ${tag}_photo128 photo128 21463824
bash tools/gpu_round.sh ${tag}_photo128 photo128
bash tools/pmc_traffic.sh ${tag}_dorn128 dorn128 19875731
bash tools/gpu_round.sh ${tag}_dorn128 dorn128
for c in res50_480 vggbn480; do
  python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --alt-steps 0 > gpurun_out/bench_${tag}_$c.json 2>/dev/null
done
python bench.py --config dorn128 --compute bf16 --steps 10 --warmup 3 --no-cpu-baseline --alt-steps 0 > gpurun_out/bench_${tag}_dorn128_bf16.json 2>/dev/null
bash tools/strong_scaling_1gpu.sh gpurun_out/strong_$tag > /dev/null 2>&1
cp gpurun_out/strong_$tag/summary.txt gpurun_out/strong_${tag}.txt
