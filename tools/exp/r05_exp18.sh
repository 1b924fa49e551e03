#!/bin/bash
mkdir -p gpurun_out
exec < /dev/null
python -m pytest tests -m gpu -x -q > gpurun_out/r05_exp18_tests.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu\|Warning\|warnings" gpurun_out/r05_exp18_tests.txt | tail -60
