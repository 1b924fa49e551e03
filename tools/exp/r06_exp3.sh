#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r06_exp3.txt
: > $O
for d in 4 65540 131076 196612 65572 4; do
  echo "== DN_WINO_DBG=$d" >> $O
  DN_WINO8=1 timeout 200 python tools/wino_timing.py $d 2>&1 | grep "loop" | sed 's/prologue.*loop/loop/; s/epilogue.*//' >> $O
done
