#!/usr/bin/env python3
"""Host-side floor of one training step: the same launch sequence on a tiny input (2 x 32 x 64), where the GPU work is
negligible, so ms/step ~ Python + ctypes + launch overhead of the ~300 launches.  usage: python tools/host_overhead.py"""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import cProfile, pstats
import torch
import bench
import supervised_dispnet_amd.loss_functions as LF
import supervised_dispnet_amd.models as models
from supervised_dispnet_amd.functional import reciprocal
from supervised_dispnet_amd.optim import FusedAdam

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
bench._quiet_init(net)
net.to(dev).train()
opt = FusedAdam(net._hot_parameters(), lr=1e-4, production_order=net._grad_production_order())
img, gt = bench.synthetic_batch(2, 32, 64, dev, 0)
gt = gt + 1.0


def step():
    depth = [reciprocal(d) for d in net(img)]
    loss = LF.l1_loss(gt, depth, "kitti")
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 50
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms/step, incl. drain %.3f ms/step" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
if "--profile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        step()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
