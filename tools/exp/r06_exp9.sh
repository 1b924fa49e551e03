#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r06_exp9.txt
: > $O
echo "== 16-wave (DN_WINO16=1), block phases" >> $O
DN_WINO16=1 timeout 200 python tools/wino_timing.py 4 2>&1 | grep -v amdgpu.ids | grep -v "epilogue: output\|kernel span" >> $O
echo "== 8-wave" >> $O
DN_WINO8=1 timeout 200 python tools/wino_timing.py 4 2>&1 | grep -v amdgpu.ids | grep -v "epilogue: output\|kernel span" >> $O
