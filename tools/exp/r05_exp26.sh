#!/bin/bash
# TIMING ONLY (results are wrong): how much the per-layer weight-gradient fences cost -- every k-th kept
mkdir -p gpurun_out
exec < /dev/null
{
for rep in 1 2; do
for k in 0 2 4 1000; do
  for b in 4 8; do
    echo "b$b keep every $k"
    DN_EXP_SKIP_FENCE=$k python bench.py --batch $b --steps 40 --warmup 10 --no-cpu-baseline --alt-steps 0 --tape-verify 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'], j['config']['launch'][:50])"
  done
done
done
} > gpurun_out/r05_exp26.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp26.txt | tail -40
