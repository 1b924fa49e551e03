#!/bin/bash
mkdir -p gpurun_out
{
P='import json,sys; l=[json.loads(x) for x in sys.stdin.read().splitlines() if x.startswith("{")][-1]; print(l["value"], l["ms_per_step"], l["ms_per_step_median"])'
B="--steps 60 --warmup 8 --no-cpu-baseline --profile-steps 0 --alt-steps 0"
for b in 4 8; do
for t in 256 128 64 32; do
echo "b$b wg_target $t"; DN_WINO_WG_TARGET=$t python bench.py --batch $b $B 2>/dev/null < /dev/null | python -c "$P"
done
echo "b$b one side stream"; DN_WGRAD_STREAMS1=1 python bench.py --batch $b $B 2>/dev/null < /dev/null | python -c "$P"
done
} > gpurun_out/r05_exp7.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp7.txt
