#!/bin/bash
# round 6, batch 7: config 4 with igemm_conv_x3b_kernel's 128 x 128 tile only above N tiles (below: 64-row tiles of igemm_conv_x3_kernel)
mkdir -p gpurun_out
O=gpurun_out/r06_exp7.txt
: > $O
for t in 192 320 448 640 192 448; do
  echo "== DN_X3B_MIN_TILES=$t" >> $O
  DN_X3B_MIN_TILES=$t python bench.py --config res50_480 --steps 10 --warmup 3 --no-cpu-baseline --alt-steps 0 --profile-steps 0 --extras 0 2>/dev/null | python -c "import sys,json; l=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); print('%.1f img/s  %.3f ms' % (l['value'], l['ms_per_step']))" >> $O
done
