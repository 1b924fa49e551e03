python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_fullsize_configs.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do python bench.py --config res50_480 --steps 10 --warmup 3 --no-cpu-baseline --alt-steps 0 --profile-steps 0 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('res50_480', round(l['value'],1), round(l['ms_per_step'],3))"; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --alt-steps 0 --profile-steps 0 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('vggbn128', round(l['value'],1), round(l['ms_per_step'],3))"
