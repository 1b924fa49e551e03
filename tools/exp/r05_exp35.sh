#!/bin/bash
# more hardware queues (GPU_MAX_HW_QUEUES, default 4) with two / three weight-gradient side streams
mkdir -p gpurun_out
exec < /dev/null
{
for rep in 1 2; do
for hq in 4 8; do
for ws in 2 3; do
  for b in 4 8; do
    echo "b$b hwq $hq wgrad-streams $ws"
    GPU_MAX_HW_QUEUES=$hq python bench.py --batch $b --wgrad-streams $ws --steps 40 --warmup 10 --no-cpu-baseline --alt-steps 0 --profile-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'])"
  done
done
done
done
} > gpurun_out/r05_exp35.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp35.txt | tail -40
