#!/usr/bin/env python3
"""Soak: N training steps of Disp_vgg_BN through the launch tape against the same N steps issued eagerly (both with the device-side
Adam counter), from the same initial state on changing batches -- parameters must stay bit-identical.  Exercises the self-resetting
K-split counters, the two side streams and the private-pool replay a few hundred times.  usage: python tools/soak_tape.py [batch] [steps]"""
import copy, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
import bench
import supervised_dispnet_amd.loss_functions as LF
import supervised_dispnet_amd.models as models
from supervised_dispnet_amd.functional import reciprocal
from supervised_dispnet_amd.graph import TapedStep, backward
from supervised_dispnet_amd.optim import FusedAdam

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")


def make(sd0=None):
    torch.manual_seed(0)
    net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
    bench._quiet_init(net)
    if sd0 is not None:
        net.load_state_dict(sd0)
    net.to(dev).train()
    opt = FusedAdam(net._hot_parameters(), lr=1e-4, production_order=net._grad_production_order())
    opt.capturable(True)
    return net, opt


batches = [bench.synthetic_batch(batch, 128, 416, dev, s) for s in range(8)]
net_e, opt_e = make()
sd0 = copy.deepcopy({k: v.detach().cpu().clone() for k, v in net_e.state_dict().items()})
net_t, opt_t = make(sd0)
img, gt = batches[0][0].clone(), batches[0][1].clone()


def step_t():
    depth = [reciprocal(d) for d in net_t(img)]
    loss = LF.l1_loss(gt, depth, "kitti")
    opt_t.zero_grad(); backward(loss); opt_t.step()
    return loss


def step_e(x, y):
    depth = [reciprocal(d) for d in net_e(x)]
    loss = LF.l1_loss(y, depth, "kitti")
    opt_e.zero_grad(); backward(loss); opt_e.step()
    return loss


ts = TapedStep(step_t, optimizer=opt_t, warmup=0, static_inputs=(img, gt))
bad = 0
for it in range(steps):
    x, y = batches[it % len(batches)]
    img.copy_(x); gt.copy_(y)
    lt = ts()
    le = step_e(x, y)
    if it % 50 == 49 or it == steps - 1:
        torch.cuda.synchronize()
        same = torch.equal(opt_t.arena.flat_p, opt_e.arena.flat_p) and torch.equal(lt, le)
        print("step %4d  loss %.6f / %.6f  parameters identical: %s" % (it + 1, float(lt), float(le), same), flush=True)
        bad += 0 if same else 1
print("SOAK", "OK" if bad == 0 else "FAILED", "batch", batch, "steps", steps)
sys.exit(0 if bad == 0 else 1)
