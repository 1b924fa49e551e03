"""Diagnostic (not a test): per-parameter gradient error of a zoo net vs the fp64 oracle, Winograd on / off.
usage: python tests/gpu_diag_zoo.py TAG"""
import os, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from oracle import detgen, losses as OL
import supervised_dispnet_amd.loss_functions as LF
import supervised_dispnet_amd.models as models
from supervised_dispnet_amd import _lib
from supervised_dispnet_amd.functional import reciprocal
from tests.cases import zoo_cases

tag = sys.argv[1]
DEV = torch.device("cuda:0")
_, cls, kwargs, ds, run = [c for c in zoo_cases() if c[0] == tag][0]
b, h, w = 2, 64, 96
x = detgen.image_batch(b, h, w, "zoo:%s:x" % tag)
gt = detgen.sparse_depth(b, h, w, "zoo:%s:gt" % tag, density=0.6, lo=0.3, hi=11.0)


def oparams(sd, dbl):
    out = {}
    for k, v in sd.items():
        v = v.detach().cpu().clone()
        if dbl and torch.is_floating_point(v):
            v = v.double()
        if torch.is_floating_point(v) and "running" not in k:
            v.requires_grad_(True)
        out[k] = v
    return out


for mode in ("wino", "direct"):
    if mode == "direct":
        os.environ["DN_NO_WINOGRAD"] = "1"
    _lib.load().dn_reload_knobs()
    net = getattr(models, cls)(**kwargs)
    detgen.fill_state_dict(net.state_dict(), "zoo:" + tag)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    net.to(DEV).train()
    depth = [reciprocal(d) for d in net(x.to(DEV))]
    (LF.l1_loss(gt.to(DEV), depth, ds) + 0.1 * LF.smooth_loss(depth)).backward()
    o32, o64 = oparams(sd0, False), oparams(sd0, True)
    d32 = [1 / d for d in run(o32, x, True)]
    (OL.l1_loss(gt, d32, ds) + 0.1 * OL.smooth_loss(d32)).backward()
    d64 = [1 / d for d in run(o64, x.double(), True)]
    (OL.l1_loss(gt.double(), d64, ds) + 0.1 * OL.smooth_loss(d64)).backward()
    rows = []
    for name, p in net.named_parameters():
        if p.grad is None or o64[name].grad is None:
            continue
        g64 = o64[name].grad
        rel = lambda a: float((a.detach().double().cpu() - g64).norm() / (g64.norm() + 1e-30))
        rows.append((rel(p.grad), rel(o32[name].grad), name))
    rows.sort(reverse=True)
    print("==", tag, mode)
    for e_hip, e_cpu, name in rows[:8]:
        print("  %-40s HIP %.3g  CPU-fp32 %.3g" % (name, e_hip, e_cpu))
