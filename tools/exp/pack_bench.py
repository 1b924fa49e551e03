#!/usr/bin/env python3
"""Time the batched weight re-lay of the metric's network in isolation (HIP events, nothing else on the device)."""
import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import torch
import supervised_dispnet_amd.loss_functions as LF
import supervised_dispnet_amd.models as models
from supervised_dispnet_amd import engine
from supervised_dispnet_amd.functional import reciprocal
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
net.init_weights(use_pretrained_weights=False)
net.to(dev).train()
x = torch.randn(4, 3, 128, 416, device=dev)
gt = torch.rand(4, 128, 416, device=dev) * 80
for _ in range(2):
    loss = LF.l1_loss(gt, [reciprocal(d) for d in net(x)], "kitti")
    loss.backward()
    engine.bump_param_epoch()
torch.cuda.synchronize()
engine.PACK_ON_SIDE_STREAM = False
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        engine.bump_param_epoch()
        engine.prepack_all(dev)
    e1.record()
    torch.cuda.synchronize()
    print("batched re-lay of every packed weight: %.1f us per step (%d table rows)" % (1e3 * e0.elapsed_time(e1) / n, len(engine.pack_table(dev).rows)))
