#!/bin/bash
# round 6, batch 8: the 16-wave Winograd form (DN_WINO16=1) against the 8-wave kernel: per-layer times + parity
mkdir -p gpurun_out
O=gpurun_out/r06_exp8.txt
: > $O
for w in 0 1 0 1; do
  echo "== DN_WINO16=$w plain" >> $O
  DN_WINO16=$w timeout 300 python tools/conv_microbench.py --reps 30 --layers c256_256_32x104,c512_512_16x52,c128_128_64x208,c128_256_32x104,c64_128_64x208 --what fwd,dgrad 2>&1 | grep -v amdgpu.ids >> $O
  echo "== DN_WINO16=$w affine+stats" >> $O
  DN_WINO16=$w timeout 300 python tools/conv_microbench.py --reps 30 --layers c256_256_32x104,c512_512_16x52,c128_128_64x208 --what fwd --affine --stats 2>&1 | grep -v amdgpu.ids >> $O
done
echo "== parity (DN_WINO16=1)" >> $O
DN_WINO16=1 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_f32x3_fp64.py tests/test_gpu_metric_shape.py -q -x 2>&1 | tail -8 >> $O
