"""Diagnostic (not a test): per-parameter gradient error of the HIP Disp_res_50 vs the fp32 and fp64 CPU oracle."""
import sys
import torch
sys.path.insert(0, ".")
import supervised_dispnet_amd.loss_functions as LF
import supervised_dispnet_amd.models as models
from oracle import detgen, losses as OL, nets_res
from supervised_dispnet_amd.functional import reciprocal

DEV = torch.device("cuda:0")
b, h, w = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (2, 64, 96)))
net = models.Disp_res_50(datasets="nyu")
detgen.fill_state_dict(net.state_dict(), "res50")
sd0 = {k: v.clone() for k, v in net.state_dict().items()}
net.to(DEV).train()
x = detgen.image_batch(b, h, w, "res50:x")
gt = detgen.sparse_depth(b, h, w, "res50:gt", density=0.6, lo=0.3, hi=11.0)
disps = net(x.to(DEV))
depth = [reciprocal(d) for d in disps]
(LF.l1_loss(gt.to(DEV), depth, "nyu") + 0.1 * LF.smooth_loss(depth)).backward()


def run(dtype):
    sd = {}
    for k, v in sd0.items():
        v = v.clone().to(dtype) if torch.is_floating_point(v) else v.clone()
        if torch.is_floating_point(v) and "running" not in k:
            v.requires_grad_(True)
        sd[k] = v
    d = nets_res.disp_res_50(sd, x.to(dtype), training=True, datasets="nyu")
    dep = [1 / t for t in d]
    (OL.l1_loss(gt.to(dtype), dep, "nyu") + 0.1 * OL.smooth_loss(dep)).backward()
    return sd


s32, s64 = run(torch.float32), run(torch.float64)
rel = lambda a, b_: float((a.double() - b_.double()).norm() / (b_.double().norm() + 1e-30))
print("%-40s %10s %10s %10s" % ("param", "hip-cpu32", "hip-f64", "cpu32-f64"))
for n, p in net.named_parameters():
    if p.grad is None:
        continue
    print("%-40s %10.3g %10.3g %10.3g" % (n, rel(p.grad.cpu(), s32[n].grad), rel(p.grad.cpu(), s64[n].grad), rel(s32[n].grad, s64[n].grad)))
