#!/bin/bash
mkdir -p gpurun_out
{
echo "== tests (tape / graph / two streams / two ranks / cli tape)"
python -m pytest tests/test_gpu_graph.py tests/test_gpu_fullsize.py tests/test_gpu_two_ranks.py tests/test_gpu_rccl.py -q -x 2>&1 < /dev/null | tail -4
python -m pytest tests/test_gpu_cli.py -q -x -k "tape or unsupervised" 2>&1 < /dev/null | tail -3
P='import json,sys; l=[json.loads(x) for x in sys.stdin.read().splitlines() if x.startswith("{")][-1]; print(l["value"], l["ms_per_step"], l["ms_per_step_median"])'
B="--steps 60 --warmup 8 --no-cpu-baseline --profile-steps 0 --alt-steps 0"
for rep in 1 2; do
for b in 4 8 32; do
echo "b$b defer"; python bench.py --batch $b $B 2>/dev/null < /dev/null | python -c "$P"
echo "b$b no-defer"; python bench.py --batch $b $B --no-defer-pack 2>/dev/null < /dev/null | python -c "$P"
done
done
echo "photo128"; python bench.py --config photo128 --steps 10 --warmup 3 --no-cpu-baseline --alt-steps 0 --profile-steps 0 2>/dev/null < /dev/null | python -c "$P"
echo "photo128 no-defer"; python bench.py --config photo128 --steps 10 --warmup 3 --no-cpu-baseline --alt-steps 0 --profile-steps 0 --no-defer-pack 2>/dev/null < /dev/null | python -c "$P"
} > gpurun_out/r05_exp9.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp9.txt
