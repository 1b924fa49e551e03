#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r06_exp4.txt
: > $O
for d in 0 262144 274432 0 262144; do
  echo "== DN_WINO8_VAR=$d plain" >> $O
  DN_WINO8_VAR=$d timeout 300 python tools/conv_microbench.py --reps 30 --layers c256_256_32x104,c512_512_16x52,c128_128_64x208 --what fwd,dgrad 2>&1 | grep -v amdgpu.ids >> $O
  echo "== DN_WINO8_VAR=$d affine+stats" >> $O
  DN_WINO8_VAR=$d timeout 300 python tools/conv_microbench.py --reps 30 --layers c256_256_32x104,c512_512_16x52 --what fwd --affine --stats 2>&1 | grep -v amdgpu.ids >> $O
done
echo "== phases 262144" >> $O
sed -i 's/os.environ\["DN_WINO_DBG"\] = "2052"/os.environ["DN_WINO_DBG"] = os.environ.get("PH_DBG", "2052")/' tools/wino8_phases.py
PH_DBG=264196 timeout 300 python tools/wino8_phases.py 2>&1 | grep -v amdgpu.ids >> $O
echo "== parity" >> $O
DN_WINO8_VAR=262144 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_f32x3_fp64.py -q -k "wino or conv_family or bn_backward_sums or f32x3" 2>&1 | tail -3 >> $O
