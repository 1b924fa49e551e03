#!/bin/bash
mkdir -p gpurun_out
exec < /dev/null
{
python tools/conv_microbench.py --batch 16 --layers p1024_256_30x40,p256_1024_30x40,p64_256_120x160,p256_64_120x160,p512_128_60x80,p128_512_60x80 --what fwd,dgrad,wgrad
BATCH=16 bash tools/pmc_micro.sh r05_x3b p1024_256_30x40,p256_1024_30x40 fwd
BATCH=16 bash tools/pmc_micro.sh r05_x3b_k64 p64_256_120x160 fwd
} > gpurun_out/r05_exp15.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp15.txt | tail -80
