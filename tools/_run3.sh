for b in 4 32; do
python bench.py --batch $b --steps 20 --warmup 4 --no-cpu-baseline --profile-steps 0 --alt-steps 0 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($b, round(l['value'],1), round(l['ms_per_step'],3), l['config'].get('tape_verified'), l['config'].get('launch')[:30], l['config'].get('graph_fallback'), l['config'].get('adam'))"
done
