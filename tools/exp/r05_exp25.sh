#!/bin/bash
# bench.py: replays follow each other on the tape's stream (lazy join, default) vs a join with the caller's stream after every step
mkdir -p gpurun_out
exec < /dev/null
{
for rep in 1 2 3; do
for opt in "" "--tape-join-every-step"; do
  for b in 4 8 32; do
    echo "b$b $opt"
    python bench.py --batch $b --steps 40 --warmup 10 --no-cpu-baseline --alt-steps 0 $opt 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'], j['ms_per_step_median'], j['config']['tape_verified'], j['config']['final_loss'])"
  done
done
done
} > gpurun_out/r05_exp25.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp25.txt | tail -60
