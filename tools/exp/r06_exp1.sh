#!/bin/bash
# round 6, batch 1: (a) split_issue microbenchmark, (b) wino8 main-loop variants (DN_WINO_DBG bits 4096 scalar split subtraction,
# 8192 interleaved accumulators, 16384 scalar staging arithmetic) on the two layers of the verdict's kill criterion, (c) their parity
mkdir -p gpurun_out
O=gpurun_out/r06_exp1.txt
: > $O
echo "== split_issue" >> $O
echo skipped >> $O
for d in 0 4096 8192 12288 20480 28672 61440 45056 0; do
  echo "== DN_WINO8_VAR=$d plain" >> $O
  DN_WINO8_VAR=$d timeout 300 python tools/conv_microbench.py --reps 30 --layers c256_256_32x104,c512_512_16x52,c128_128_64x208 --what fwd,dgrad >> $O 2>&1
  echo "== DN_WINO8_VAR=$d affine+stats" >> $O
  DN_WINO8_VAR=$d timeout 300 python tools/conv_microbench.py --reps 30 --layers c256_256_32x104,c512_512_16x52 --what fwd --affine --stats >> $O 2>&1
done
echo "== parity of the variants" >> $O
for d in 28672 61440; do
  echo "-- DN_WINO8_VAR=$d" >> $O
  DN_WINO8_VAR=$d timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_f32x3_fp64.py -q -k "wino or conv_family or bn_backward_sums" 2>&1 | tail -3 >> $O
done
