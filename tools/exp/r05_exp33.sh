#!/bin/bash
mkdir -p gpurun_out
exec < /dev/null
{
for rep in 1 2 3 4; do
for ov in 0 1; do
    echo "b32 adam-overlap $ov"
    python bench.py --batch 32 --adam-overlap $ov --steps 40 --warmup 10 --no-cpu-baseline --alt-steps 0 --profile-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'], j['config']['launch'][:60])"
done
done
} > gpurun_out/r05_exp33.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp33.txt | tail -20
