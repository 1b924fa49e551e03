#!/bin/bash
# launch-tape fences riding on the stop event of the launch in front of them (default) vs event records of their own (DN_NO_RIDING_FENCES=1)
mkdir -p gpurun_out
exec < /dev/null
{
python -m pytest tests/test_gpu_graph.py tests/test_gpu_two_ranks.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -3
for rep in 1 2 3; do
for nr in 0 1; do
  for b in 4 8 32; do
    echo "b$b no-riding $nr"
    if [ $nr = 1 ]; then export DN_NO_RIDING_FENCES=1; else unset DN_NO_RIDING_FENCES; fi
    python bench.py --batch $b --steps 40 --warmup 10 --no-cpu-baseline --alt-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'], j['config']['launch'][:95], j['config']['tape_verified'][:20], j['config']['final_loss'])"
  done
done
done
} > gpurun_out/r05_exp29.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp29.txt | tail -60
