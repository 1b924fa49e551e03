#!/bin/bash
mkdir -p gpurun_out
exec < /dev/null
{
python tools/exp/pack_bench.py
python -m pytest tests/test_gpu_kernels.py -q -x -k "winograd or wino or conv_family" 2>&1 | tail -2
python -m pytest tests/test_gpu_models.py -q -x -k "vgg_bn_forward_backward or compute_mode or config5 or dorn" 2>&1 | tail -2
P='import json,sys; l=[json.loads(x) for x in sys.stdin.read().splitlines() if x.startswith("{")][-1]; print(l["value"], l["ms_per_step"], l["ms_per_step_median"])'
B="--steps 60 --warmup 8 --no-cpu-baseline --profile-steps 0 --alt-steps 0"
for b in 4 8 32; do echo "b$b"; python bench.py --batch $b $B 2>/dev/null | python -c "$P"; done
python bench.py --config dorn128 --compute bf16 --steps 10 --warmup 3 --no-cpu-baseline --alt-steps 0 --profile-steps 0 2>/dev/null | python -c "$P"
} > gpurun_out/r05_exp13.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp13.txt | tail -20
