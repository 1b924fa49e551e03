#!/bin/bash
# weight gradients of small layers held back to share a fence: group size x size threshold
mkdir -p gpurun_out
exec < /dev/null
{
python -m pytest tests/test_gpu_graph.py tests/test_gpu_models.py -m gpu -x -q -k "tape or two_streams or vgg_bn or parity" 2>&1 | grep -E "passed|failed" | tail -3
for rep in 1 2; do
for cfg in "1 6" "2 6" "3 6" "4 6" "2 20" "3 20" "2 3"; do
  set -- $cfg
  for b in 4 8; do
    echo "b$b group $1 max-gflop $2"
    python bench.py --batch $b --steps 40 --warmup 10 --no-cpu-baseline --alt-steps 0 --wgrad-fence-group $1 --wgrad-defer-gflop $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'], j['config']['launch'][:50], j['config']['tape_verified'][:20], j['config']['final_loss'])"
  done
done
done
for cfg in "1 6" "2 6"; do
  set -- $cfg
  echo "b32 group $1 max-gflop $2"
  python bench.py --batch 32 --steps 30 --warmup 10 --no-cpu-baseline --alt-steps 0 --wgrad-fence-group $1 --wgrad-defer-gflop $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'], j['config']['launch'][:50], j['config']['tape_verified'][:20], j['config']['final_loss'])"
done
} > gpurun_out/r05_exp27.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp27.txt | tail -70
