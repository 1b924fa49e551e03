#!/bin/bash
# what the pending BatchNorm + ReLU in the loader and the BatchNorm statistics in the epilogue cost the encoder's forward convolutions
mkdir -p gpurun_out
exec < /dev/null
{
L=c64_64_128x416,c128_128_64x208,c256_256_32x104,c512_512_16x52,c512_512_8x26
for rep in 1 2; do
for opt in "" "--affine" "--stats" "--affine --stats"; do
  echo "== opts: $opt"
  python tools/conv_microbench.py --batch 32 --layers $L --what fwd $opt
done
done
} > gpurun_out/r05_exp19.txt 2>&1
grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r05_exp19.txt | tail -120
