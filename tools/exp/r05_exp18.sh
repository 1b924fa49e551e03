#!/bin/bash
mkdir -p gpurun_out
exec < /dev/null
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 400 --durations=8 > gpurun_out/r05_exp18_tests.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu\|Warning\|warnings" gpurun_out/r05_exp18_tests.txt | tail -40
