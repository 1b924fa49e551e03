#!/bin/bash
mkdir -p gpurun_out
{
echo "== last-arrival probe"; python tools/exp/last_arrival_probe.py
echo "== new tests"
python -m pytest tests/test_gpu_metric_shape.py -q -x 2>&1 | tail -15
python -m pytest tests/test_gpu_cli.py -q -k "legacy or compute or evaluation_chain or dorn" 2>&1 | tail -15
python -m pytest tests/test_gpu_models.py -q -k "reciprocal" 2>&1 | tail -5
B="--steps 40 --warmup 8 --no-cpu-baseline --profile-steps 0 --alt-steps 0"
P='import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l["value"], l["ms_per_step"], l["ms_per_step_median"])'
for b in 4 8; do
echo "== step b$b default"; python bench.py --batch $b $B 2>/dev/null | python -c "$P"
echo "== step b$b split 208/512"; DN_WINO_SPLITK_MAXBLOCKS=208 DN_WINO_SPLITK_TARGET=512 python bench.py --batch $b $B 2>/dev/null | python -c "$P"
echo "== step b$b split 256/512"; DN_WINO_SPLITK_MAXBLOCKS=256 DN_WINO_SPLITK_TARGET=512 python bench.py --batch $b $B 2>/dev/null | python -c "$P"
echo "== step b$b split 208/512 minch 4"; DN_WINO_SPLITK_MAXBLOCKS=208 DN_WINO_SPLITK_TARGET=512 DN_WINO_SPLITK_MINCH=4 python bench.py --batch $b $B 2>/dev/null | python -c "$P"
done
echo "== res50_480 per layer"
python bench.py --config res50_480 --steps 10 --warmup 3 --per-layer --no-cpu-baseline --alt-steps 0 > gpurun_out/r05_exp2_res50.json 2> gpurun_out/r05_exp2_res50_per_layer.txt
python -c "$P" < gpurun_out/r05_exp2_res50.json
echo "== default bench with the new cpu baseline"
python bench.py > gpurun_out/r05_exp2_bench.json 2> gpurun_out/r05_exp2_bench.err
python -c 'import json,sys; l=json.loads(open("gpurun_out/r05_exp2_bench.json").read().strip().splitlines()[-1]); print(l["value"], l["ms_per_step"]); print(json.dumps(l["cpu_baseline"], indent=1))'
} > gpurun_out/r05_exp2.txt 2>&1
tail -30 gpurun_out/r05_exp2.txt
