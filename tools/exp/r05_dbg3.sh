#!/bin/bash
mkdir -p gpurun_out
exec < /dev/null
{
for c in "1 1" "0 1" "1 0" "0 0"; do
  timeout 150 python tools/exp/r05_dbg3.py $c 2>&1 | grep "step"
  echo "rc $?"
done
} > gpurun_out/r05_dbg3.txt 2>&1
cat gpurun_out/r05_dbg3.txt
