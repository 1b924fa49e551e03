python -m pytest tests/test_gpu_graph.py -x -q -m gpu 2>&1 | tail -15
bash tools/_sweep.sh "4 8 32" "DN_BENCH=1"
for l in eager tape; do for b in 4 8; do
python bench.py --batch $b --steps 40 --warmup 8 --no-cpu-baseline --profile-steps 0 --alt-steps 0 --launch $l 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', $b, l['value'], l['ms_per_step'], l['ms_per_step_median'], l['config'].get('launch'), l['config'].get('graph_fallback'))"
done; done
