#!/usr/bin/env python3
"""Host-side floor of one training step: the same launch sequence on a tiny input (2 x 32 x 64), where the GPU work is
negligible, so ms/step ~ Python + ctypes + launch overhead of the ~220 launches.  usage: python tools/host_overhead.py [--shape N H W] [--profile]"""
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
shape = (2, 32, 64)
if "--shape" in sys.argv:
    i = sys.argv.index("--shape")
    shape = tuple(int(v) for v in sys.argv[i + 1:i + 4])
img, gt = bench.synthetic_batch(shape[0], shape[1], shape[2], dev, 0)
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
if "--floor" in sys.argv:
    # GPU-side floor of the step: block the GPU with a long spin kernel, enqueue whole steps behind it (the host runs ahead), and time
    # the steps with events -- what the launch sequence costs when no launch ever waits for the host
    for rep in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(int(60e-3 * 2.1e9))
        th0 = time.perf_counter()
        step()
        e0.record()
        for _ in range(4):
            step()
        e1.record()
        th1 = time.perf_counter()
        torch.cuda.synchronize()
        print("pre-queued: GPU %.3f ms/step (host enqueued 5 steps in %.1f ms)" % (e0.elapsed_time(e1) / 4, (th1 - th0) * 1e3))
if "--profile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        step()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
