#!/usr/bin/env python3
"""Which ATen / runtime operations (copies, fills, elementwise kernels) still run inside one bench step, and who calls them.
usage (GPU box): python tools/aten_on_hot_path.py"""
import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
import supervised_dispnet_amd.loss_functions as LF
import supervised_dispnet_amd.models as models
import supervised_dispnet_amd.utils as U
from supervised_dispnet_amd.functional import reciprocal
from supervised_dispnet_amd.optim import FusedAdam

dev = torch.device("cuda:0")
step, opt, state, desc = bench.build_workload("vggbn128", 32, dev, 0, models, LF, U, reciprocal, FusedAdam)
for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_stack_n=6)
rows = [e for e in ev if e.key.startswith("aten::") or "Memcpy" in e.key or "Memset" in e.key or "copyBuffer" in e.key or "fill" in e.key.lower()]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    print("%-40s count/2 steps %4d  self cuda %.1f us" % (e.key[:40], e.count, getattr(e, "self_device_time_total", 0.0)))
    for s in (e.stack or [])[:5]:
        if "supervised_dispnet_amd" in s or "bench.py" in s:
            print("      ", s[-110:])
