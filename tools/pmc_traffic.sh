#!/bin/bash
# HBM traffic per kernel launch from the rocprofv3 PMC counters (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage (GPU box): bash tools/pmc_traffic.sh <tag> [config] [arena elements]
#   -> gpurun_out/pmc_<tag>/{FETCH_SIZE,WRITE_SIZE}/..., gpurun_out/pmc_<tag>_traffic.json
tag=${1:-x}
cfg=${2:-vggbn128}
arena=${3:-19873156}
R=$(pwd)
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  DN_WGRAD_STREAM=0 rocprofv3 --kernel-trace --pmc $c -d $out/$c -o p --output-format csv -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --profile-steps 0 --no-cpu-baseline --alt-steps 0 > $out/$c.log 2>&1 || echo "pass $c failed"
done
cd $R
python tools/pmc_traffic_summary.py $out $(python -c "print($arena * 4.0)") > gpurun_out/pmc_${tag}_traffic.json
