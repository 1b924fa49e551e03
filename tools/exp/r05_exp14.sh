#!/bin/bash
mkdir -p gpurun_out
exec < /dev/null
{
free -g | head -2
python -m pytest tests/test_gpu_metric_shape.py -q -x -k "config3 or config4" 2>&1 | grep -v "Warning\|warnings.warn" | tail -30
} > gpurun_out/r05_exp14.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp14.txt | tail -30
