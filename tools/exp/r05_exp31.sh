#!/bin/bash
# live fences through dn_stream_fence (default) vs torch's wait_stream (--torch-fences): eager launches, and the tape for reference
mkdir -p gpurun_out
exec < /dev/null
{
timeout 600 python -m pytest tests/test_gpu_graph.py tests/test_gpu_kernels.py -m gpu -x -q --timeout 200 2>&1 | grep -E "passed|failed" | tail -3
for rep in 1 2 3; do
for opt in "" "--torch-fences"; do
  for cfg in "4 eager" "8 eager" "32 eager" "4 tape"; do
    set -- $cfg
    echo "b$1 $2 $opt"
    python bench.py --batch $1 --launch $2 --steps 40 --warmup 10 --no-cpu-baseline --alt-steps 0 --profile-steps 0 $opt 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'], j['config']['final_loss'])"
  done
done
done
} > gpurun_out/r05_exp31.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp31.txt | tail -60
