#!/bin/bash
mkdir -p gpurun_out
{
P='import json,sys; l=[json.loads(x) for x in sys.stdin.read().splitlines() if x.startswith("{")][-1]; print(l["value"], l["ms_per_step"], l["ms_per_step_median"])'
B="--steps 60 --warmup 8 --no-cpu-baseline --profile-steps 0 --alt-steps 0"
for b in 4 8 16 32; do
echo "b$b rule"; python bench.py --batch $b $B 2>/dev/null < /dev/null | python -c "$P"
echo "b$b target 256"; DN_WINO_WG_TARGET=256 python bench.py --batch $b $B 2>/dev/null < /dev/null | python -c "$P"
echo "b$b target 128"; DN_WINO_WG_TARGET=128 python bench.py --batch $b $B 2>/dev/null < /dev/null | python -c "$P"
done
for b in 4 8; do
echo "b$b rule, 1 side stream"; python bench.py --batch $b $B --wgrad-streams 1 2>/dev/null < /dev/null | python -c "$P"
echo "b$b target 192"; DN_WINO_WG_TARGET=192 python bench.py --batch $b $B 2>/dev/null < /dev/null | python -c "$P"
echo "b$b target 96"; DN_WINO_WG_TARGET=96 python bench.py --batch $b $B 2>/dev/null < /dev/null | python -c "$P"
done
} > gpurun_out/r05_exp8.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp8.txt
