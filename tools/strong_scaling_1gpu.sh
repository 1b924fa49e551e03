#!/bin/bash
# One-GPU bound of the literal metric (global b32 split over N ranks -> b = 32/N per GPU): img/s at b4/b8/b16/b32 for the ways the host
# can issue the step: launch tape (default), eager launches, hipGraph replay -- and (round 5) the launch tape with the DATA-PARALLEL
# machinery live on a single-rank own-RCCL communicator (--reducer rccl1: gradient buckets, the tape cut at every bucket, event fences,
# ncclAllReduce on the library's stream; only the wire is missing), plain and with the HBM stand-in of a ring all-reduce
# (--comm-standin 1: a PROJECTION).
# usage: tools/strong_scaling_1gpu.sh OUTDIR [modes]      modes default: "tape eager graph rccl1 rccl1+standin"
out=${1:-gpurun_out/strong}
modes=${2:-"tape eager graph rccl1 rccl1+standin"}
mkdir -p $out
for l in $modes; do
  for b in 4 8 16 32; do
    case $l in
      rccl1) flags="--launch tape --reducer rccl1" ;;
      rccl1+standin) flags="--launch tape --reducer rccl1 --comm-standin 1" ;;
      *) flags="--launch $l" ;;
    esac
    python bench.py --batch $b --steps 40 --warmup 8 --no-cpu-baseline $flags --profile-steps 0 --alt-steps 0 > $out/b${b}_$l.json 2> $out/b${b}_$l.err
    python - <<PY
import json
l = [json.loads(x) for x in open("$out/b${b}_$l.json").read().splitlines() if x.startswith("{")][-1]
c = l["config"]
print("b=%2d launch=%-14s %8.1f img/s  %.3f ms/step (median %.3f)  %s | comm %s, %s buckets | %s" % ($b, "$l", l["value"], l["ms_per_step"], l["ms_per_step_median"], c.get("graph_fallback"), c.get("comm"), c.get("comm_buckets"), c.get("launch")))
PY
  done
done | tee $out/summary.txt
