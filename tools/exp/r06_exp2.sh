#!/bin/bash
# round 6, batch 2: where a wave of the 8-wave Winograd kernel spends a chunk (phase timestamps) + the round-4 ablations re-taken
mkdir -p gpurun_out
O=gpurun_out/r06_exp2.txt
: > $O
timeout 300 python tools/wino8_phases.py >> $O 2>&1
echo "== ablations" >> $O
timeout 600 bash tools/wino8_ablate.sh >> $O 2>&1
