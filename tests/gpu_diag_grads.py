"""Diagnostic (not a pytest): per-parameter gradient error of the HIP path and of PyTorch-CPU fp32, both against an fp64
run of the oracle, for the tiny Disp_vgg_BN case.  python tests/gpu_diag_grads.py > gpurun_out/diag_grads.txt"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
import supervised_dispnet_amd.loss_functions as LF
import supervised_dispnet_amd.models as models
from oracle import detgen, losses as OL, nets as ON
from supervised_dispnet_amd.functional import reciprocal

DEV = torch.device("cuda:0")
b, h, w = (2, 64, 96) if len(sys.argv) < 2 else (2, 128, 416)
tag = "vggbn_tiny" if len(sys.argv) < 2 else "vggbn_cfg"
net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
detgen.fill_state_dict(net.state_dict(), "vggbn")
sd0 = {k: v.clone() for k, v in net.state_dict().items()}
net.to(DEV).train()
x = detgen.image_batch(b, h, w, tag + ":x")
gt = detgen.sparse_depth(b, h, w, tag + ":gt", density=0.3 if len(sys.argv) < 2 else 0.05)
disps = net(x.to(DEV))
depth = [reciprocal(d) for d in disps]
(LF.l1_loss(gt.to(DEV), depth, "kitti") + 0.1 * LF.smooth_loss(depth)).backward()


def run(dtype):
    sd = {k: (v.detach().clone().to(dtype) if torch.is_floating_point(v) else v.clone()) for k, v in sd0.items()}
    for k, v in sd.items():
        if torch.is_floating_point(v) and "running" not in k:
            v.requires_grad_(True)
    d = ON.disp_vgg_bn(sd, x.to(dtype), training=True)
    dep = [1 / t for t in d]
    (OL.l1_loss(gt.to(dtype), dep, "kitti") + 0.1 * OL.smooth_loss(dep)).backward()
    return sd, d


s32, d32 = run(torch.float32)
s64, d64 = run(torch.float64)
for i in range(4):
    sc = float(d64[i].detach().abs().max())
    print("disp%d  hip-vs-f64 %.3e   cpu32-vs-f64 %.3e (rel to max)" % (
        i, float((disps[i].detach().cpu().double() - d64[i].detach()).abs().max()) / sc, float((d32[i].detach().double() - d64[i].detach()).abs().max()) / sc))
print("%-36s %10s %10s %10s %8s" % ("param", "max|g64|", "hip/f64", "cpu32/f64", "ratio"))
for name, p in net.named_parameters():
    g64 = s64[name].grad
    if g64 is None:
        continue
    sc = float(g64.abs().max()) + 1e-30
    eh = float((p.grad.cpu().double() - g64).abs().max()) / sc
    ec = float((s32[name].grad.double() - g64).abs().max()) / sc
    print("%-36s %10.3e %10.3e %10.3e %8.2f" % (name, sc, eh, ec, eh / max(ec, 1e-12)))

# ---- flip hypothesis: are the bad layers' errors concentrated in single channels, and do those channels have a
# pre-ReLU value within fp32 round-off of zero in the fp64 run?
import torch.nn.functional as F
rec = []
orig_relu = F.relu
def spy(t, *a, **k):
    v = t.detach()
    am = v.abs()
    idx = int(am.argmin())
    c = (idx // (v.shape[2] * v.shape[3])) % v.shape[1]
    rec.append((tuple(v.shape), float(am.min()), float(am.max()), c, int((am < 1e-6 * am.max()).sum()), int((am < 1e-5 * am.max()).sum())))
    return orig_relu(t, *a, **k)
ON.F.relu = spy
with torch.no_grad():
    ON.disp_vgg_bn({k: v.detach() for k, v in s64.items()}, x.double(), training=True)
ON.F.relu = orig_relu
print("\nfp64 pre-ReLU inputs per BN layer: shape, min|z|, max|z|, channel of min, #(|z|<1e-6 max), #(|z|<1e-5 max)")
for i, r in enumerate(rec):
    print(i, r)
for name in ("features.features.28.bias", "features.features.28.weight", "features.features.31.bias", "features.features.25.bias"):
    g64 = s64[name].grad
    e = (dict(net.named_parameters())[name].grad.cpu().double() - g64).abs()
    top = torch.topk(e, 4)
    print(name, "top err channels", top.indices.tolist(), ["%.2e" % v for v in top.values.tolist()], "median err %.2e" % float(e.median()))
