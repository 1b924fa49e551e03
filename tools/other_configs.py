#!/usr/bin/env python3
"""Secondary configurations (BASELINE.json configs[3]): img/s of a training step of Disp_res_50 at 480x640 b16 and of
Disp_vgg_BN at 480x640 b16 -- not the headline metric, a sanity / regression number for the same kernels at other shapes."""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
import supervised_dispnet_amd.loss_functions as LF
import supervised_dispnet_amd.models as models
from supervised_dispnet_amd.functional import reciprocal
from supervised_dispnet_amd.optim import FusedAdam
import bench

dev = torch.device("cuda:0")
for name, ctor, ds, B, H, W in (("Disp_res_50", models.Disp_res_50, "nyu", 16, 480, 640),
                                ("Disp_vgg_BN", lambda datasets: models.Disp_vgg_BN(datasets=datasets, with_classifier=False), "nyu", 16, 480, 640)):
    torch.manual_seed(0)
    net = ctor(datasets=ds)
    bench._quiet_init(net)
    net.to(dev).train()
    params = net._hot_parameters() if hasattr(net, "_hot_parameters") else [p for p in net.parameters()]
    order = net._grad_production_order() if hasattr(net, "_grad_production_order") else None
    opt = FusedAdam(params, lr=1e-4, betas=(0.9, 0.999), production_order=order) if order is not None else torch.optim.Adam(params, lr=1e-4)
    g = torch.Generator().manual_seed(0)
    img = ((torch.rand(B, 3, H, W, generator=g) - 0.5) / 0.5).to(dev)
    gt = (torch.rand(B, H, W, generator=g) * 9.5 + 0.5).to(dev)

    def step():
        disps = net(img)
        loss = LF.l1_loss(gt, [reciprocal(d) for d in disps], ds)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 8
    for _ in range(n):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%-12s %dx%d b%d: %.1f ms/step  %.1f img/s  loss %.5f  peak mem %.1f GB" % (name, H, W, B, dt * 1e3, B / dt, float(loss), torch.cuda.max_memory_allocated() / 2**30))
    del net, opt
    torch.cuda.empty_cache()
