#!/bin/bash
mkdir -p gpurun_out
exec < /dev/null
{
for rep in 1 2 3; do
for nm in 0 1 2 3; do
  for b in 32 4 8; do
    echo "b$b nmajor $nm"
    DN_WINO_NMAJOR=$nm python bench.py --batch $b --steps 40 --warmup 10 --no-cpu-baseline --alt-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'], j['roofline'].get('achieved'))"
  done
done
done
} > gpurun_out/r05_exp17.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp17.txt | tail -120
