#!/bin/bash
mkdir -p gpurun_out
exec < /dev/null
{
python tools/soak_fold.py 4 400
python tools/soak_fold.py 8 200
python tools/soak_fold.py 32 60
python tools/soak_tape.py 4 200 2>&1 | tail -3
} > gpurun_out/r05_exp12.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp12.txt | tail -20
