#!/bin/bash
mkdir -p gpurun_out
exec < /dev/null
{
P='import json,sys; l=[json.loads(x) for x in sys.stdin.read().splitlines() if x.startswith("{")][-1]; print(l["value"], l["ms_per_step"], l["ms_per_step_median"])'
B="--steps 60 --warmup 8 --no-cpu-baseline --profile-steps 0 --alt-steps 0"
for b in 4 8 32; do
for pb in 512 256 128 64 32; do
echo "b$b pack_blocks $pb"; DN_PACK_BLOCKS=$pb python bench.py --batch $b $B 2>/dev/null | python -c "$P"
done
done
} > gpurun_out/r05_exp11.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp11.txt
