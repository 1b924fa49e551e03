#!/bin/bash
# Adam per bucket under the backward pass (--adam-overlap 1) vs one pass after it (0) at 8 / 16 / 32 images; one vs two weight-gradient streams at 16
mkdir -p gpurun_out
exec < /dev/null
{
for rep in 1 2 3; do
for ov in 0 1; do
  for b in 8 16 32; do
    echo "b$b adam-overlap $ov"
    python bench.py --batch $b --adam-overlap $ov --steps 40 --warmup 10 --no-cpu-baseline --alt-steps 0 --profile-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'], j['config']['adam'])"
  done
done
for ws in 1 2; do
  echo "b16 wgrad-streams $ws"
  python bench.py --batch 16 --wgrad-streams $ws --steps 40 --warmup 10 --no-cpu-baseline --alt-steps 0 --profile-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'])"
done
done
} > gpurun_out/r05_exp32.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp32.txt | tail -60
