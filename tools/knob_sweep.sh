#!/bin/bash
# One bench line per (batch, environment setting): A/B of DN_* knobs on ONE box (box-to-box spread of a build is +-2 %).
# usage: bash tools/knob_sweep.sh "<batches>" "ENV1=a ENV2=b" "ENV1=c" ...   (one bench line per batch x setting)
bs=$1; shift
for e in "$@"; do for b in $bs; do
  env $e python bench.py --batch $b --steps 40 --warmup 8 --no-cpu-baseline --profile-steps 0 --alt-steps 0 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s b=%2d %8.1f img/s  %.3f ms (median %.3f)' % ('$e', $b, l['value'], l['ms_per_step'], l['ms_per_step_median']))"
done; done
