#!/bin/bash
# round 5, experiment 1: small-grid policies of the Winograd forward / input gradient at 4 and 8 images
mkdir -p gpurun_out
{
python tools/exp/fullsplit_check.py
L=c512_512_16x52,c256_512_16x52,c256_256_32x104,c128_256_32x104,c512_512_8x26,c1024_512_8x26,c768_256_16x52,c384_128_32x104
for b in 4 8; do
echo "== default b$b"; python tools/conv_microbench.py --batch $b --layers $L --what fwd,dgrad --affine
echo "== 4-wave split up to 208 blocks b$b"; DN_WINO_SPLITK_MAXBLOCKS=208 DN_WINO_SPLITK_TARGET=512 python tools/conv_microbench.py --batch $b --layers $L --what fwd,dgrad --affine
echo "== wino8 full split b$b"; DN_WINO8_FULLSPLIT=1 python tools/conv_microbench.py --batch $b --layers $L --what fwd,dgrad --affine
echo "== wino8 full split minch 8 b$b"; DN_WINO8_FULLSPLIT=1 DN_WINO8_FULLSPLIT_MINCH=8 python tools/conv_microbench.py --batch $b --layers $L --what fwd,dgrad --affine
done
B="--steps 40 --warmup 8 --no-cpu-baseline --profile-steps 0 --alt-steps 0"
for b in 4 8; do
echo "== step b$b default"; python bench.py --batch $b $B 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['ms_per_step_median'])"
echo "== step b$b 4-wave split 208"; DN_WINO_SPLITK_MAXBLOCKS=208 DN_WINO_SPLITK_TARGET=512 python bench.py --batch $b $B 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['ms_per_step_median'])"
echo "== step b$b wino8 full split"; DN_WINO8_FULLSPLIT=1 python bench.py --batch $b $B 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['ms_per_step_median'])"
done
} > gpurun_out/r05_exp1.txt 2>&1
tail -5 gpurun_out/r05_exp1.txt
