export DN_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 2 8; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) bench.py --gpus $n --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 > gpurun_out/r03f/bench_n$n.json 2> gpurun_out/r03f/bench_n$n.err
echo "rc=$?"
python - <<PY
import json
try:
    l=json.loads(open("gpurun_out/r03f/bench_n$n.json").read().strip().splitlines()[-1])
    print($n, round(l["value"],1), round(l["ms_per_step"],3), l["scaling"], l["config"].get("launch"), l["config"].get("comm"), l["config"].get("adam"), l["config"].get("graph_fallback"), "weak:", (l.get("weak_scaling") or {}).get("value"))
except Exception as e:
    print("ERR", e); print(open("gpurun_out/r03f/bench_n$n.err").read()[-1500:])
PY
done
