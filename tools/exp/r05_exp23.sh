#!/bin/bash
mkdir -p gpurun_out
exec < /dev/null
{
for b in 4 8 32; do
  python bench.py --batch $b --steps 40 --warmup 10 --no-cpu-baseline --alt-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'], 'host replay ms', j['config']['tape_replay_host_ms'])"
done
} > gpurun_out/r05_exp23.txt 2>&1
cat gpurun_out/r05_exp23.txt | grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl"
