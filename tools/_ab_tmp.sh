python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "conv_family" 2>&1 | tail -15
python -m pytest tests/test_gpu_models.py -q -m gpu -x 2>&1 | tail -5
for c in vggbn128 res50_480 photo128; do
for v in "DN_X=1" "DN_NO_X3_WGRAD=1"; do
env $v python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --alt-steps 0 --profile-steps 0 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $v', round(l['ms_per_step'],3), round(l['value'],1))"
done; done
python bench.py --config res50_480 --steps 5 --warmup 2 --no-cpu-baseline --alt-steps 0 --per-layer 2> gpurun_out/res50_layers.err >/dev/null
grep -E "wgrad" gpurun_out/res50_layers.err | cut -c1-160 | head -70
