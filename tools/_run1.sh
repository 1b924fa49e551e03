mkdir -p gpurun_out/r03f
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "split or fused_into" 2>&1 | tail -5 > gpurun_out/r03f/t_split.log
cat gpurun_out/r03f/t_split.log
for b in 4 8; do
for e in "DN_NO_WINO_SPLITK=1" "DN_X=1"; do
  env $e python bench.py --batch $b --steps 40 --warmup 8 --no-cpu-baseline --profile-steps 0 --alt-steps 0 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', $b, l['value'], l['ms_per_step'], l['ms_per_step_median'])"
done; done | tee gpurun_out/r03f/strong_split.txt
