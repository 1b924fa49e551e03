#!/bin/bash
# One-GPU bound of the literal metric (global b32 split over N ranks -> b = 32/N per GPU): img/s at b4/b8/b16/b32 for the three ways
# the host can issue the step: launch tape (default), eager launches, hipGraph replay.
# usage: tools/strong_scaling_1gpu.sh OUTDIR
out=${1:-gpurun_out/strong}
mkdir -p $out
for l in tape eager graph; do
  for b in 4 8 16 32; do
    python bench.py --batch $b --steps 40 --warmup 8 --no-cpu-baseline --launch $l --profile-steps 0 --alt-steps 0 > $out/b${b}_$l.json 2> $out/b${b}_$l.err
    python - <<PY
import json
l = json.loads(open("$out/b${b}_$l.json").read().strip().splitlines()[-1])
print("b=%2d launch=%-5s  %8.1f img/s  %.3f ms/step (median %.3f)  %s" % ($b, "$l", l["value"], l["ms_per_step"], l["ms_per_step_median"], l["config"].get("graph_fallback")))
PY
  done
done | tee $out/summary.txt
