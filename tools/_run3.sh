python -m pytest tests/test_gpu_graph.py tests/test_gpu_kernels.py -x -q -m gpu -k "bn or pool or graph" 2>&1 | tail -3
for e in DN_X=1 DN_X=2; do
env $e python bench.py --steps 10 --warmup 3 --no-cpu-baseline --alt-steps 0 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=l['roofline_hbm']['by_kernel']
print('$e', round(l['ms_per_step'],3), {k.split('::')[-1][:34]:(round(v['GBps']),round(v['ms_per_step'],3)) for k,v in h.items() if 'Pool' in k or 'apply' in k})"
done
bash tools/_sweep.sh "4" DN_X=1
