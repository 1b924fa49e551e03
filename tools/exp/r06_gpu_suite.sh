#!/bin/bash
# full GPU suite + the driver's default bench call (with the round-6 extras) on one box
mkdir -p gpurun_out
tag=${1:-a}
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r06_${tag}_gpu_tests.txt
( time timeout 1500 python bench.py ) > gpurun_out/r06_${tag}_bench.json 2> gpurun_out/r06_${tag}_bench.err
tail -3 gpurun_out/r06_${tag}_bench.err
