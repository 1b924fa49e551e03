# quick loop for Winograd weight-gradient / forward edits: conv parity tests, fp64 pins, then the layer microbenchmark (single stream)
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_f32x3_fp64.py -x -q -m gpu 2>&1 | tail -25
DN_WGRAD_STREAM=0 python tools/conv_microbench.py --layers c64_64_128x416,c128_128_64x208,c256_256_32x104,c512_512_16x52,c512_512_8x26 --what ${WHAT:-wgrad} --affine 2>&1 | grep -v amdgpu.ids | tail -12
