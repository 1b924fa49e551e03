#!/bin/bash
mkdir -p gpurun_out
exec < /dev/null
for b in 4 32; do
  python bench.py --batch $b --steps 20 --warmup 10 --no-cpu-baseline --alt-steps 0 --profile-steps 0 --tape-host-profile 2>&1 >/dev/null | grep "tape-host-profile"
done > gpurun_out/r05_exp24.txt 2>&1
cat gpurun_out/r05_exp24.txt
