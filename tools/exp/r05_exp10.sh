#!/bin/bash
mkdir -p gpurun_out
exec < /dev/null
{
python -m pytest tests/test_gpu_models.py -q -k "every_surviving_switch" 2>&1 | grep -v "Warning\|warnings.warn\|^$" | tail -40
} > gpurun_out/r05_exp10.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp10.txt | tail -40
