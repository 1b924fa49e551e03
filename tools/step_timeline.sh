#!/bin/bash
# rocprofv3 kernel trace of a bench run at batch $1 -> per-kernel totals + one step as a timeline with queue ids (tools/step_timeline.py).
out=${2:-gpurun_out/timeline}
mkdir -p $out
B=${1:-4}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b4; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_b4 -- python $GRAFT_REPO_ROOT/bench.py --batch $B --steps 30 --warmup 8 --no-cpu-baseline --profile-steps 0 --alt-steps 0 > /tmp/b4.json 2>/tmp/b4.err
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_b4 -name '*kernel_trace.csv' | head -1)
python tools/trace_gaps.py $f 0.5 > $out/b${B}_trace_summary.txt
python tools/step_timeline.py $f 0 > $out/b${B}_timeline.txt
python tools/step_gaps.py $f 3 | tail -20 > $out/b${B}_step_gaps.txt
tail -4 $out/b${B}_timeline.txt
