#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r06_exp5.txt
: > $O
echo "== wino_timing 8-wave, plain" >> $O
DN_WINO8=1 timeout 200 python tools/wino_timing.py 4 2>&1 | grep -v amdgpu.ids >> $O
echo "== wino_timing 8-wave, bn stats" >> $O
DN_WINO8=1 timeout 200 python tools/wino_timing.py 4 bn 2>&1 | grep -v amdgpu.ids >> $O
echo "== wino_timing 4-wave (DN_WINO8=0)" >> $O
DN_WINO8=0 timeout 200 python tools/wino_timing.py 4 2>&1 | grep -v amdgpu.ids >> $O
echo "== nt weights" >> $O
for d in 0 1048576 0 1048576; do
  echo "-- DN_WINO8_VAR=$d" >> $O
  DN_WINO8_VAR=$d timeout 300 python tools/conv_microbench.py --reps 30 --layers c256_256_32x104,c512_512_16x52,c128_128_64x208 --what fwd,dgrad 2>&1 | grep -v amdgpu.ids >> $O
done
echo "== split_issue" >> $O
timeout 120 tools/ubench/bin/split_issue >> $O 2>&1
