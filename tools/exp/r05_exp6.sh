#!/bin/bash
mkdir -p gpurun_out
{
P='import json,sys; l=[json.loads(x) for x in sys.stdin.read().splitlines() if x.startswith("{")][-1]; print(l["value"], l["ms_per_step"], l["ms_per_step_median"], l["config"]["launch"][:40])'
B="--steps 60 --warmup 8 --no-cpu-baseline --profile-steps 0 --alt-steps 0"
for rep in 1 2; do
for b in 4 8 32; do
echo "b$b fold"; python bench.py --batch $b $B 2>/dev/null < /dev/null | python -c "$P"
echo "b$b no-fold"; python bench.py --batch $b $B --no-fold 2>/dev/null < /dev/null | python -c "$P"
done
done
} > gpurun_out/r05_exp6.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp6.txt
