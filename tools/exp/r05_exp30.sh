#!/bin/bash
# fused masked-loss forward + borrowed seed gradient (default) vs --no-loss-fusion
mkdir -p gpurun_out
exec < /dev/null
{
for rep in 1 2 3; do
for opt in "" "--no-loss-fusion"; do
  for b in 4 8 32; do
    echo "b$b $opt"
    python bench.py --batch $b --steps 40 --warmup 10 --no-cpu-baseline --alt-steps 0 $opt 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'], j['config']['launch'][:40], j['config']['tape_verified'][:20], j['config']['final_loss'])"
  done
done
done
} > gpurun_out/r05_exp30.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp30.txt | tail -60
