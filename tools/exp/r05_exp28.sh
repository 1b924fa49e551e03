#!/bin/bash
# tile order (DN_WINO_NMAJOR) in the other compute modes and configs
mkdir -p gpurun_out
exec < /dev/null
{
for rep in 1 2; do
for nm in 0 1; do
  for cfg in "dorn128 bf16" "vggbn128 bf16" "vggbn128 f32" "res50_480 f32x3" "photo128 f32x3"; do
    set -- $cfg
    echo "$1 $2 nmajor $nm"
    DN_WINO_NMAJOR=$nm python bench.py --config $1 --compute $2 --steps 12 --warmup 4 --no-cpu-baseline --alt-steps 0 --profile-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'])"
  done
done
done
} > gpurun_out/r05_exp28.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp28.txt | tail -60
