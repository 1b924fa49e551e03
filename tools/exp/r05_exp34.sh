#!/bin/bash
# two vs three vs four weight-gradient side streams at 4 / 8 / 16 images
mkdir -p gpurun_out
exec < /dev/null
{
for rep in 1 2; do
for ws in 2 3 4; do
  for b in 4 8 16; do
    echo "b$b wgrad-streams $ws"
    python bench.py --batch $b --wgrad-streams $ws --steps 40 --warmup 10 --no-cpu-baseline --alt-steps 0 --profile-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'])"
  done
done
done
} > gpurun_out/r05_exp34.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp34.txt | tail -40
