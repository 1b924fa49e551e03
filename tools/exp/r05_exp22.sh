#!/bin/bash
mkdir -p gpurun_out
exec < /dev/null
R=$(pwd)
for B in 4; do
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_g; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_g -- python $R/bench.py --batch $B --steps 30 --warmup 8 --no-cpu-baseline --profile-steps 0 --alt-steps 0 > /tmp/g.json 2>/tmp/g.err
cd $R
f=$(find /tmp/prof_g -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py $f 0 > gpurun_out/r05_exp22_b${B}_median_timeline.txt
python tools/step_gaps.py $f 4 | tail -12
done
