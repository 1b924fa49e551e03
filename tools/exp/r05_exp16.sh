#!/bin/bash
# Winograd forward/dgrad tile order within an XCD: N fastest (default) vs tile-row fastest (DN_WINO_NMAJOR=1: one weight slice per XCD)
mkdir -p gpurun_out
exec < /dev/null
{
L=c512_512_16x52,c512_512_32x104,c256_256_32x104,c256_256_64x208,c128_128_64x208,c128_128_128x416,c64_64_128x416,c512_512_8x26,c256_512_16x52
for nm in 0 1 0 1; do
  echo "== nmajor $nm"
  DN_WINO_NMAJOR=$nm python tools/conv_microbench.py --batch 32 --layers $L --what fwd,dgrad
done
for rep in 1 2; do
for nm in 0 1; do
  for b in 32 4; do
    echo "b$b nmajor $nm"
    DN_WINO_NMAJOR=$nm python bench.py --batch $b --steps 40 --warmup 10 --no-cpu-baseline --alt-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'], j['roofline'].get('achieved'))"
  done
done
done
} > gpurun_out/r05_exp16.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp16.txt | tail -120
