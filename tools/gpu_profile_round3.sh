#!/bin/bash
# Round-3 profile set in one GPU call: tests, bench line (+ per layer), rocprofv3 kernel stats, PMC traffic, other configs, one-GPU strong scaling.
# usage (GPU box): bash tools/gpu_profile_round3.sh <tag>
tag=${1:-r03_a}
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/tests_$tag.log 2>&1; tail -2 gpurun_out/tests_$tag.log
bash tools/pmc_traffic.sh $tag
bash tools/gpu_round.sh $tag
for c in dorn128 photo128 res50_480 vggbn480; do
  python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --alt-steps 0 > gpurun_out/bench_${tag}_$c.json 2>/dev/null
done
python bench.py --config dorn128 --compute bf16 --steps 10 --warmup 3 --no-cpu-baseline --alt-steps 0 > gpurun_out/bench_${tag}_dorn128_bf16.json 2>/dev/null
bash tools/strong_scaling_1gpu.sh gpurun_out/strong_$tag > /dev/null 2>&1
cp gpurun_out/strong_$tag/summary.txt gpurun_out/strong_${tag}.txt
