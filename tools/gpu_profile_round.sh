#!/bin/bash
# One GPU call that produces everything profiles/ needs for a round: tests, bench line, rocprofv3 kernel stats, PMC traffic.
# usage (GPU box): bash tools/gpu_profile_round.sh <tag>
tag=${1:-x}
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/tests_$tag.log 2>&1; tail -2 gpurun_out/tests_$tag.log
bash tools/pmc_traffic.sh $tag
bash tools/gpu_round.sh $tag
