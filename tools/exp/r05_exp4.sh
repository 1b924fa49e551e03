#!/bin/bash
# round 5, GPU call 4: validation of the round's changes + the numbers asked for under "Next round" items 1, 3, 4
mkdir -p gpurun_out
{
echo "== tests (cli legacy/compute, metric shape, graph/bn, zoo res50)"
python -m pytest tests/test_gpu_cli.py -q -x -k "legacy or compute" 2>&1 < /dev/null | tail -3
python -m pytest tests/test_gpu_graph.py tests/test_gpu_fullsize_configs.py -q -x -k "bn_backward or config4 or tape" 2>&1 < /dev/null | tail -3
echo "== pack kernel A/B (isolated)"
python tools/exp/pack_bench.py < /dev/null
DN_PACK_V1=1 python tools/exp/pack_bench.py < /dev/null
P='import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l["value"], l["ms_per_step"], l["ms_per_step_median"])'
echo "== res50_480 (hoisted bn_bwd_apply)"
python bench.py --config res50_480 --steps 10 --warmup 3 --per-layer --no-cpu-baseline --alt-steps 0 > gpurun_out/r05_exp4_res50.json 2> gpurun_out/r05_exp4_res50_per_layer.txt < /dev/null
python -c "$P" < gpurun_out/r05_exp4_res50.json
grep "bn_bwd_apply" gpurun_out/r05_exp4_res50_per_layer.txt
echo "== strong scaling incl. the data-parallel machinery on a single-rank communicator"
bash tools/strong_scaling_1gpu.sh gpurun_out/strong_r05_a "tape rccl1 rccl1+standin" < /dev/null
echo "== b4 / b32 pack A/B in the step"
B="--steps 40 --warmup 8 --no-cpu-baseline --profile-steps 0 --alt-steps 0"
for b in 4 32; do
echo "b$b new"; python bench.py --batch $b $B 2>/dev/null < /dev/null | python -c "$P"
echo "b$b old pack"; DN_PACK_V1=1 python bench.py --batch $b $B 2>/dev/null < /dev/null | python -c "$P"
done
} > gpurun_out/r05_exp4.txt 2>&1
tail -40 gpurun_out/r05_exp4.txt
