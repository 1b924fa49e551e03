#!/bin/bash
mkdir -p gpurun_out
{
echo "== legacy test"; python -m pytest tests/test_gpu_cli.py -q -x -k "legacy" 2>&1 | grep -v "Warning\|pin_memory" | tail -30
echo "== pack kernel: winograd tests"; python -m pytest tests/test_gpu_kernels.py -q -x -k "winograd or wino or conv_family" 2>&1 | tail -5
python -m pytest tests/test_gpu_models.py -q -x -k "vgg_bn_forward_backward or compute_mode" 2>&1 | tail -3
B="--steps 40 --warmup 8 --no-cpu-baseline --profile-steps 0 --alt-steps 0"
P='import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l["value"], l["ms_per_step"], l["ms_per_step_median"])'
for b in 4 32; do
echo "== step b$b (new pack, split 208/512)"; DN_WINO_SPLITK_MAXBLOCKS=208 DN_WINO_SPLITK_TARGET=512 python bench.py --batch $b $B 2>/dev/null | python -c "$P"
done
echo "== res50_480 per layer"
python bench.py --config res50_480 --steps 10 --warmup 3 --per-layer --no-cpu-baseline --alt-steps 0 > gpurun_out/r05_exp3_res50.json 2> gpurun_out/r05_exp3_res50_per_layer.txt
python -c "$P" < gpurun_out/r05_exp3_res50.json
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r05_exp3_b4 -o b4 --output-format csv -- python $R/bench.py --batch 4 --steps 20 --warmup 4 --profile-steps 0 --no-cpu-baseline --alt-steps 0 > $R/gpurun_out/prof_r05_exp3_b4.log 2>&1
cd $R; ls gpurun_out/prof_r05_exp3_b4/*/ | head
f=$(ls gpurun_out/prof_r05_exp3_b4/*/*kernel_stats.csv | head -1); head -40 $f
} > gpurun_out/r05_exp3.txt 2>&1
tail -5 gpurun_out/r05_exp3.txt
