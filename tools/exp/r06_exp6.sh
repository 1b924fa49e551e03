#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r06_exp6.txt
: > $O
for pz in 0 1 0 1; do
  echo "== DN_WINO8_PERSIST=$pz plain" >> $O
  DN_WINO8_PERSIST=$pz timeout 300 python tools/conv_microbench.py --reps 30 --layers c256_256_32x104,c512_512_16x52,c128_128_64x208,c64_128_64x208,c128_256_32x104 --what fwd,dgrad 2>&1 | grep -v amdgpu.ids >> $O
  echo "== DN_WINO8_PERSIST=$pz affine+stats" >> $O
  DN_WINO8_PERSIST=$pz timeout 300 python tools/conv_microbench.py --reps 30 --layers c256_256_32x104,c512_512_16x52,c128_128_64x208 --what fwd --affine --stats 2>&1 | grep -v amdgpu.ids >> $O
done
echo "== parity (persistent on)" >> $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_f32x3_fp64.py tests/test_gpu_metric_shape.py -q 2>&1 | tail -5 >> $O
