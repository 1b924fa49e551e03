python -m pytest tests/test_gpu_graph.py tests/test_gpu_two_ranks.py -x -q -m gpu 2>&1 | grep -v "^\[W\|amdgpu.ids\|Gloo\|RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl" | tail -3
for ov in 0 1 0 1; do for b in 4 8 32; do
python bench.py --batch $b --steps 40 --warmup 8 --no-cpu-baseline --profile-steps 0 --alt-steps 0 --adam-overlap $ov 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap=$ov', $b, round(l['value'],1), round(l['ms_per_step'],3), round(l['ms_per_step_median'],3), l['config'].get('graph_fallback'))"
done; done
