#!/usr/bin/env python3
"""Per-kernel time of one Disp_res_50 training step at 480x640 b16 (BASELINE configs[3] shape) via engine.PROFILE events."""
import pathlib, sys, collections
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
import supervised_dispnet_amd.loss_functions as LF
import supervised_dispnet_amd.models as models
from supervised_dispnet_amd import engine
from supervised_dispnet_amd.functional import reciprocal
import bench
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = models.Disp_res_50(datasets="nyu")
bench._quiet_init(net)
net.to(dev).train()
opt = torch.optim.Adam([p for p in net.parameters()], lr=1e-4)
g = torch.Generator().manual_seed(0)
img = ((torch.rand(16, 3, 480, 640, generator=g) - 0.5) / 0.5).to(dev)
gt = (torch.rand(16, 480, 640, generator=g) * 9.5 + 0.5).to(dev)
def step():
    loss = LF.l1_loss(gt, [reciprocal(d) for d in net(img)], "nyu")
    opt.zero_grad(); loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
engine.PROFILE = []
step()
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for name, flops, e0, e1, tag, _nbytes, _ab in engine.PROFILE:
    a = agg[name]; a[0] += flops; a[1] += e0.elapsed_time(e1); a[2] += 1
tot = sum(v[1] for v in agg.values())
print("conv-family total %.2f ms" % tot)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-50s %7.2f ms  %6.1f TF  x%d" % (k, v[1], v[0] / v[1] / 1e9, v[2]))
