#!/bin/bash
mkdir -p gpurun_out
{
echo "== fold test"; python -m pytest tests/test_gpu_graph.py -q -x -k "folded" 2>&1 < /dev/null | tail -15
echo "== related tests"; python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_graph.py tests/test_gpu_two_ranks.py tests/test_gpu_rccl.py -q -x 2>&1 < /dev/null | tail -6
P='import json,sys; l=[json.loads(x) for x in sys.stdin.read().splitlines() if x.startswith("{")][-1]; print(l["value"], l["ms_per_step"], l["ms_per_step_median"], l["config"]["launch"][:40])'
B="--steps 40 --warmup 8 --no-cpu-baseline --profile-steps 0 --alt-steps 0"
for b in 4 8 32; do
echo "b$b"; python bench.py --batch $b $B 2>/dev/null < /dev/null | python -c "$P"
done
} > gpurun_out/r05_exp5.txt 2>&1
tail -30 gpurun_out/r05_exp5.txt
