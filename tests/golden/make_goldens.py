#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

Imports /root/reference through a small shim loader (the container lacks torchvision, path.py,
scipy.misc.imresize/imread ...), feeds it closed-form inputs/weights from oracle.detgen, and stores
inputs-by-formula + outputs.  Nothing of the reference's source travels: only numbers.

    python tests/golden/make_goldens.py            # writes tests/golden/*.npz

The committed .npz files are what the CPU test-suite pins the oracle against; the GPU box never
reads /root/reference.
"""
import importlib.util
import os
import pathlib
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

REF = pathlib.Path(os.environ.get("DISPNET_REFERENCE", "/root/reference"))
HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))

from oracle import detgen  # noqa: E402


# ----------------------------------------------------------------------------- shim loader
def _install_shims():
    # torchvision: layout only (cfg "D" with BN) -- all arithmetic stays torch.nn
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")

    class _VGG(nn.Module):
        def __init__(self):
            super().__init__()
            cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
            layers, c = [], 3
            for v in cfg:
                if v == "M":
                    layers.append(nn.MaxPool2d(2, 2))
                else:
                    layers += [nn.Conv2d(c, v, 3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
                    c = v
            self.features = nn.Sequential(*layers)
            self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
            # the 123.6 M-parameter classifier is never used by the hot path; keep it tiny here
            self.classifier = nn.Sequential(nn.Linear(8, 8), nn.ReLU(True), nn.Dropout(), nn.Linear(8, 8),
                                            nn.ReLU(True), nn.Dropout(), nn.Linear(8, 8))

    tvm.vgg16_bn = lambda pretrained=False, **kw: _VGG()

    class _VGGPlain(nn.Module):                 # torchvision.models.vgg16() layout (cfg "D", no BatchNorm): Disp_vgg_feature.py:85
        def __init__(self):
            super().__init__()
            cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
            layers, c = [], 3
            for v in cfg:
                if v == "M":
                    layers.append(nn.MaxPool2d(2, 2))
                else:
                    layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU(inplace=True)]
                    c = v
            self.features = nn.Sequential(*layers)
            self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
            self.classifier = nn.Sequential(nn.Linear(8, 8), nn.ReLU(True), nn.Dropout(), nn.Linear(8, 8),
                                            nn.ReLU(True), nn.Dropout(), nn.Linear(8, 8))

    tvm.vgg16 = lambda pretrained=False, **kw: _VGGPlain()

    # ResNet: layout only (torchvision.models.ResNet with BasicBlock / Bottleneck), used by networks/resnet_encoder.py
    def _c3(i, o, s=1):
        return nn.Conv2d(i, o, 3, s, 1, bias=False)

    def _c1(i, o, s=1):
        return nn.Conv2d(i, o, 1, s, bias=False)

    class _Basic(nn.Module):
        expansion = 1

        def __init__(self, inplanes, planes, stride=1, downsample=None):
            super().__init__()
            self.conv1, self.bn1 = _c3(inplanes, planes, stride), nn.BatchNorm2d(planes)
            self.relu = nn.ReLU(inplace=True)
            self.conv2, self.bn2 = _c3(planes, planes), nn.BatchNorm2d(planes)
            self.downsample = downsample

        def forward(self, x):
            idt = x if self.downsample is None else self.downsample(x)
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.bn2(self.conv2(out))
            return self.relu(out + idt)

    class _Bottle(nn.Module):
        expansion = 4

        def __init__(self, inplanes, planes, stride=1, downsample=None):
            super().__init__()
            self.conv1, self.bn1 = _c1(inplanes, planes), nn.BatchNorm2d(planes)
            self.conv2, self.bn2 = _c3(planes, planes, stride), nn.BatchNorm2d(planes)
            self.conv3, self.bn3 = _c1(planes, planes * 4), nn.BatchNorm2d(planes * 4)
            self.relu = nn.ReLU(inplace=True)
            self.downsample = downsample

        def forward(self, x):
            idt = x if self.downsample is None else self.downsample(x)
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.relu(self.bn2(self.conv2(out)))
            out = self.bn3(self.conv3(out))
            return self.relu(out + idt)

    class _ResNet(nn.Module):
        def __init__(self, block, layers, num_classes=1000):
            super().__init__()
            self.inplanes = 64
            self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
            self.bn1 = nn.BatchNorm2d(64)
            self.relu = nn.ReLU(inplace=True)
            self.maxpool = nn.MaxPool2d(3, 2, 1)
            self.layer1 = self._make_layer(block, 64, layers[0])
            self.layer2 = self._make_layer(block, 128, layers[1], 2)
            self.layer3 = self._make_layer(block, 256, layers[2], 2)
            self.layer4 = self._make_layer(block, 512, layers[3], 2)
            self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
            self.fc = nn.Linear(512 * block.expansion, num_classes)

        def _make_layer(self, block, planes, blocks, stride=1):
            ds = None
            if stride != 1 or self.inplanes != planes * block.expansion:
                ds = nn.Sequential(_c1(self.inplanes, planes * block.expansion, stride), nn.BatchNorm2d(planes * block.expansion))
            layers = [block(self.inplanes, planes, stride, ds)]
            self.inplanes = planes * block.expansion
            layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
            return nn.Sequential(*layers)

    tvr = types.ModuleType("torchvision.models.resnet")
    tvr.BasicBlock, tvr.Bottleneck, tvr.model_urls = _Basic, _Bottle, {}
    tvm.resnet = tvr
    tvm.ResNet = _ResNet
    tvm.resnet18 = lambda pretrained=False, **kw: _ResNet(_Basic, [2, 2, 2, 2])
    tvm.resnet34 = lambda pretrained=False, **kw: _ResNet(_Basic, [3, 4, 6, 3])
    tvm.resnet50 = lambda pretrained=False, **kw: _ResNet(_Bottle, [3, 4, 6, 3])
    tvm.resnet101 = lambda pretrained=False, **kw: _ResNet(_Bottle, [3, 4, 23, 3])
    tvm.resnet152 = lambda pretrained=False, **kw: _ResNet(_Bottle, [3, 8, 36, 3])
    sys.modules["torchvision.models.resnet"] = tvr
    tv.models = tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm
    import scipy.misc
    scipy.misc.imresize = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("imresize stub"))
    scipy.misc.imread = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("imread stub"))
    pm = types.ModuleType("path")
    pm.Path = pathlib.Path
    sys.modules["path"] = pm
    if not hasattr(np, "int"):
        np.int = int
    sys.path.insert(0, str(REF))


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, str(REF / rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _np(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    out = HERE / (name + ".npz")
    np.savez_compressed(out, **arrays)
    print("wrote", out.name, "%.1f KB" % (out.stat().st_size / 1024))


# --------------------------------------------------------------------------------- sections
def gold_dispnets(ref_dispnets):
    """Config 1: DispNetS forward on 2x(3,128,416), train and eval."""
    net = ref_dispnets.DispNetS(datasets="kitti")
    detgen.fill_state_dict(net.state_dict(), "dispnets")
    x = detgen.image_batch(2, 128, 416, "dispnets:x")
    net.train()
    outs = net(x)
    arrays = {}
    for i, o in enumerate(outs):
        for k, v in detgen.summarize(o).items():
            arrays["train%d_%s" % (i, k)] = v
    arrays["train3_full"] = _np(outs[3])          # coarsest scale in full
    loss = sum((o * detgen.uniform(tuple(o.shape), "dispnets:g%d" % i, -1, 1)).sum() for i, o in enumerate(outs))
    loss.backward()
    for key in ("conv1.0.weight", "conv7.2.weight", "upconv7.0.weight", "iconv3.0.weight", "predict_disp1.0.weight",
                "iconv1.0.bias"):
        g = dict(net.named_parameters())[key].grad
        for k, v in detgen.summarize(g, stride=53).items():
            arrays["grad:%s:%s" % (key, k)] = v
    net.eval()
    with torch.no_grad():
        e = net(x)
    for k, v in detgen.summarize(e).items():
        arrays["eval_%s" % k] = v
    save("dispnets_cfg1", **arrays)


def _vgg_case(ref_vgg, b, h, w, tag, full):
    net = ref_vgg.Disp_vgg_BN(datasets="kitti")
    detgen.fill_state_dict(net.state_dict(), "vggbn")
    x = detgen.image_batch(b, h, w, tag + ":x")
    gt = detgen.sparse_depth(b, h, w, tag + ":gt", density=0.3 if full else 0.05)
    return net, x, gt


def gold_vgg_bn(ref_vgg, ref_loss):
    """Disp_vgg_BN fwd (4 outputs), l1-loss backward grads, BN running stats; tiny (full) + config shape."""
    for tag, (b, h, w), full in (("vggbn_tiny", (2, 64, 96), True), ("vggbn_cfg", (2, 128, 416), False)):
        net, x, gt = _vgg_case(ref_vgg, b, h, w, tag, full)
        net.train()
        disps = net(x)
        depth = [1 / d for d in disps]
        loss = ref_loss.l1_loss(gt, depth, "kitti") + 0.1 * ref_loss.smooth_loss(depth)
        loss.backward()
        arrays = {"loss": np.float64(loss.item())}
        for i, o in enumerate(disps):
            if full:
                arrays["disp%d" % i] = _np(o)
            for k, v in detgen.summarize(o).items():
                arrays["disp%d_%s" % (i, k)] = v
        params = dict(net.named_parameters())
        for key in ("features.features.0.weight", "features.features.0.bias", "features.features.1.weight",
                    "features.features.1.bias", "features.features.40.weight", "features.features.41.weight",
                    "features.features.17.weight", "upconv4.0.weight", "upconv0.0.weight", "upconv0.0.bias",
                    "iconv4.0.weight", "iconv2.0.weight", "iconv0.0.weight", "iconv0.0.bias", "disp0.0.weight",
                    "disp3.0.weight", "disp2.0.bias"):
            g = params[key].grad
            if full and g.numel() <= 20000:
                arrays["grad:%s" % key] = _np(g)
            for k, v in detgen.summarize(g, stride=53).items():
                arrays["grad:%s:%s" % (key, k)] = v
        sd = net.state_dict()
        for key in ("features.features.1.running_mean", "features.features.1.running_var",
                    "features.features.41.running_mean", "features.features.41.running_var"):
            arrays["bn:%s" % key] = _np(sd[key])
        arrays["bn:nbt"] = _np(sd["features.features.1.num_batches_tracked"])
        if full:
            net.eval()
            with torch.no_grad():
                arrays["eval_disp0"] = _np(net(x))
        save(tag, **arrays)


def gold_train_step(ref_vgg, ref_loss):
    """One iteration of (Disp_vgg_BN, L1, b=2, Adam lr 1e-4) as train.py:441-522 does it."""
    net, x, gt = _vgg_case(ref_vgg, 2, 64, 96, "trainstep", True)
    net.train()
    opt = torch.optim.Adam([p for n, p in net.named_parameters() if "classifier" not in n], lr=1e-4, betas=(0.9, 0.999))
    losses = []
    for _ in range(2):
        disps = net(x)
        depth = [1 / d for d in disps]
        loss = 1.0 * ref_loss.l1_loss(gt, depth, "kitti") + 0.0 * ref_loss.smooth_loss(depth)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    arrays = {"losses": np.array(losses, dtype=np.float64)}
    sd = net.state_dict()
    for key in ("features.features.0.weight", "features.features.1.weight", "features.features.1.running_var",
                "features.features.40.bias", "upconv4.0.weight", "iconv0.0.weight", "disp0.0.weight", "disp0.0.bias"):
        for k, v in detgen.summarize(sd[key], stride=31).items():
            arrays["post:%s:%s" % (key, k)] = v
    save("trainstep_vggbn_l1", **arrays)


def gold_losses(ref_loss):
    b, h, w = 3, 32, 64
    gt = detgen.sparse_depth(b, h, w, "loss:gt", density=0.4, lo=0.5, hi=90.0)  # some > 80 -> invalid
    arrays = {}
    for ds in ("kitti", "nyu"):
        for name in ("l1_loss", "l2_loss", "berhu_loss", "Scale_invariant_loss"):
            if name == "berhu_loss" and ds == "nyu":
                continue  # NameError in the reference (loss_functions.py:148-159)
            depth = [detgen.uniform((b, 1, h >> i, w >> i), "loss:pred%d" % i, 1e-4, 95.0).requires_grad_() for i in range(4)]
            v = getattr(ref_loss, name)(gt, depth, ds)
            v.backward()
            arrays["%s:%s" % (name, ds)] = np.float64(v.item())
            arrays["%s:%s:grad" % (name, ds)] = _np(depth[0].grad)
    for name in ("Multiscale_L1_loss", "Multiscale_FULL_L1_loss", "Multiscale_L2_loss", "Multiscale_berhu_loss",
                 "Multiscale_scale_inv_loss"):
        pools = ("bilinear", "max", "avg") if name == "Multiscale_L1_loss" else (("nearest",) if "FULL" in name else (None,))
        for pool in pools:
            depth = [detgen.uniform((b, 1, h >> i, w >> i), "loss:pred%d" % i, 1e-4, 95.0).requires_grad_() for i in range(4)]
            v = getattr(ref_loss, name)(gt, depth) if pool is None else getattr(ref_loss, name)(gt, depth, pool)
            v.backward()
            key = name if pool is None else "%s:%s" % (name, pool)
            arrays[key] = np.float64(v.item())
            for i in range(4):
                arrays["%s:grad%d" % (key, i)] = _np(depth[i].grad)
    depth = [detgen.uniform((b, 1, h >> i, w >> i), "loss:pred%d" % i, 0.5, 60.0).requires_grad_() for i in range(4)]
    v = ref_loss.smooth_loss(depth)
    v.backward()
    arrays["smooth_loss"] = np.float64(v.item())
    for i in range(4):
        arrays["smooth_loss:grad%d" % i] = _np(depth[i].grad)
    p = detgen.uniform((b, 8, h, w), "loss:ord", 0.0, 1.0).requires_grad_()
    v = ref_loss.smooth_DORN_loss(p)
    v.backward()
    arrays["smooth_DORN_loss"] = np.float64(v.item())
    arrays["smooth_DORN_loss:grad"] = _np(p.grad)
    # empty-mask sample -> NaN (mean of empty)
    gt0 = gt.clone(); gt0[1] = 0
    depth = [detgen.uniform((b, 1, h, w), "loss:pred0", 1e-4, 95.0)]
    arrays["l1_loss:empty_sample"] = np.float64(ref_loss.l1_loss(gt0, depth, "kitti").item())
    # explainability
    m = [detgen.uniform((b, 2, h >> i, w >> i), "loss:mask%d" % i, 0.05, 0.95).requires_grad_() for i in range(2)]
    v = ref_loss.explainability_loss(m); v.backward()
    arrays["explainability_loss"] = np.float64(v.item())
    arrays["explainability_loss:grad0"] = _np(m[0].grad)
    save("losses", **arrays)


def gold_compute_errors(ref_loss):
    arrays = {}
    for ds, (b, h, w), hi in (("kitti", (3, 128, 416), 90.0), ("nyu", (2, 48, 64), 11.0)):
        gt = detgen.sparse_depth(b, h, w, "err:gt:" + ds, density=0.3, lo=0.5, hi=hi)
        pred = detgen.uniform((b, h, w), "err:pred:" + ds, 1e-4, hi)
        arrays["errors:%s" % ds] = np.array(ref_loss.compute_errors(gt, pred, ds), dtype=np.float64)
        arrays["errors:%s:median" % ds] = np.array(ref_loss.compute_errors(gt, pred, ds, True, True), dtype=np.float64)
    save("compute_errors", **arrays)


def warp_inputs(b, h, w, tag):
    img = detgen.uniform((b, 3, h, w), tag + ":img", -1, 1)
    depth = detgen.uniform((b, h, w), tag + ":depth", 2.0, 30.0)
    pose = detgen.uniform((b, 6), tag + ":pose", -0.05, 0.05)
    fx, fy, cx, cy = 241.67 * w / 416, 246.28 * h / 128, 204.17 * w / 416, 59.0 * h / 128
    k = torch.tensor([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=torch.float32).repeat(b, 1, 1)
    return img, depth, pose, k, torch.inverse(k)


def gold_warp(ref_warp, ref_loss):
    arrays = {}
    b, h, w = 2, 24, 40
    img, depth, pose, k, kinv = warp_inputs(b, h, w, "warp")
    arrays["kinv"] = _np(kinv)
    import torch.nn.functional as F
    orig = F.grid_sample
    for ac in (False, True):
        F.grid_sample = (lambda i, g, padding_mode="zeros", _ac=ac: orig(i, g, padding_mode=padding_mode, align_corners=_ac))
        ref_warp.F.grid_sample = F.grid_sample
        for pad in ("zeros", "border"):
            for rot in ("euler", "quat"):
                d = depth.clone().requires_grad_(); p = pose.clone().requires_grad_()
                ref_warp.pixel_coords = None
                out = ref_warp.inverse_warp(img, d, p, k, kinv, rot, pad)
                (out * detgen.uniform(tuple(out.shape), "warp:g", -1, 1)).sum().backward()
                key = "ac%d:%s:%s" % (int(ac), pad, rot)
                arrays[key] = _np(out)
                arrays[key + ":gdepth"] = _np(d.grad)
                arrays[key + ":gpose"] = _np(p.grad)
    # photometric loss at both align_corners settings (4 scales, 2 refs)
    b, h, w = 2, 32, 64
    tgt, _, _, k, kinv = warp_inputs(b, h, w, "photo")
    refs = [detgen.uniform((b, 3, h, w), "photo:ref%d" % i, -1, 1) for i in range(2)]
    pose = detgen.uniform((b, 2, 6), "photo:pose", -0.03, 0.03)
    for ac in (False, True):
        F.grid_sample = (lambda i, g, padding_mode="zeros", _ac=ac: orig(i, g, padding_mode=padding_mode, align_corners=_ac))
        ref_warp.F.grid_sample = F.grid_sample
        for with_mask in (False, True):
            ref_warp.pixel_coords = None
            depth = [detgen.uniform((b, 1, h >> i, w >> i), "photo:d%d" % i, 2.0, 30.0).requires_grad_() for i in range(4)]
            pz = pose.clone().requires_grad_()
            # training-mode PoseExpNet(output_exp=False) hands over [None]*4 (models/PoseExpNet.py:84-92)
            mask = [detgen.uniform((b, 2, h >> i, w >> i), "photo:m%d" % i, 0.1, 0.9) for i in range(4)] if with_mask else [None] * 4
            v = ref_loss.photometric_reconstruction_loss(tgt, refs, k, kinv, depth, mask, pz, "euler", "zeros")
            v.backward()
            key = "photo:ac%d:mask%d" % (int(ac), int(with_mask))
            arrays[key] = np.float64(v.item())
            arrays[key + ":gpose"] = _np(pz.grad)
            for i in range(4):
                arrays[key + ":gdepth%d" % i] = _np(depth[i].grad)
    F.grid_sample = orig
    ref_warp.F.grid_sample = orig
    # QUIRK pin: a bare None mask becomes [None] and zip() truncates the loss to scale 0 only
    ref_warp.pixel_coords = None
    depth = [detgen.uniform((b, 1, h >> i, w >> i), "photo:d%d" % i, 2.0, 30.0) for i in range(4)]
    arrays["photo:bare_none_mask"] = np.float64(
        ref_loss.photometric_reconstruction_loss(tgt, refs, k, kinv, depth, None, pose, "euler", "zeros").item())
    save("warp", **arrays)


def gold_layers(ref_layers):
    x = detgen.uniform((2, 3, 20, 28), "ssim:x", 0, 1).requires_grad_()
    y = detgen.uniform((2, 3, 20, 28), "ssim:y", 0, 1).requires_grad_()
    s = ref_layers.SSIM()(x, y)
    (s * detgen.uniform(tuple(s.shape), "ssim:g", -1, 1)).sum().backward()
    disp = detgen.uniform((2, 1, 20, 28), "esm:disp", 0.1, 5).requires_grad_()
    img = detgen.uniform((2, 3, 20, 28), "esm:img", 0, 1)
    e = ref_layers.get_smooth_loss(disp, img); e.backward()
    save("layers", ssim=_np(s), ssim_gx=_np(x.grad), ssim_gy=_np(y.grad), edge_smooth=np.float64(e.item()),
         edge_smooth_gdisp=_np(disp.grad))


def gold_dorn(ref_dorn, ref_utils, ref_loss):
    arrays = {}
    for ds, hi in (("kitti", 85.0), ("nyu", 11.0)):
        d = detgen.uniform((2, 16, 24), "sid:d:" + ds, 0.0, hi)
        for kc in (71, 80):
            lab = ref_utils.get_labels_sid(d, ordinal_c=kc, dataset=ds)
            arrays["labels:%s:%d" % (ds, kc)] = _np(lab).astype(np.int32)
            arrays["decode:%s:%d" % (ds, kc)] = _np(ref_utils.get_depth_sid(lab, ordinal_c=kc, dataset=ds))
    pre = detgen.uniform((2, 2 * 12, 10, 14), "orl:pre", -3, 3).requires_grad_()
    dec, ordc = ref_dorn.OrdinalRegressionLayer()(pre)
    arrays["orl:decode"] = _np(dec).astype(np.int64)
    arrays["orl:ord"] = _np(ordc)
    gt = detgen.sparse_depth(2, 10, 14, "orl:gt", density=0.6, lo=0.5, hi=90)
    tgt = ref_utils.get_labels_sid(gt, ordinal_c=12, dataset="kitti")
    v = ref_loss.DORN_loss(gt, ordc, tgt, "kitti"); v.backward()
    arrays["dorn_loss"] = np.float64(v.item())
    arrays["dorn_loss:gpre"] = _np(pre.grad)
    # eval-mode forward of the full DORN-head net (dropout = identity)
    net = ref_dorn.Disp_vgg_BN_DORN(datasets="kitti", ordinal_c=8)
    detgen.fill_state_dict(net.state_dict(), "vggdorn")
    net.eval()
    x = detgen.image_batch(1, 64, 96, "vggdorn:x")
    with torch.no_grad():
        dec, ordc = net(x)
    arrays["net:decode"] = _np(dec).astype(np.int64)
    arrays["net:ord"] = _np(ordc)
    save("dorn", **arrays)


def synthetic_kitti_scene(n_pts=20000):
    """Synthetic calibration + point cloud in KITTI conventions (values by formula)."""
    p_rect = np.array([7.215377e+02, 0, 6.095593e+02, 4.485728e+01, 0, 7.215377e+02, 1.728540e+02, 2.163791e-01,
                       0, 0, 1, 2.745884e-03])
    r_rect = np.array([9.999239e-01, 9.837760e-03, -7.445048e-03, -9.869795e-03, 9.999421e-01, -4.278459e-03,
                       7.402527e-03, 4.351614e-03, 9.999631e-01])
    r = np.array([7.533745e-03, -9.999714e-01, -6.166020e-04, 1.480249e-02, 7.280733e-04, -9.998902e-01,
                  9.998621e-01, 7.523790e-03, 1.480755e-02])
    t = np.array([-4.069766e-03, -7.631618e-02, -2.717806e-01])
    u = detgen.uniform((n_pts, 3), "kitti:pts", 0, 1, dtype=torch.float64).numpy()
    velo = np.zeros((n_pts, 4), dtype=np.float32)
    velo[:, 0] = (-2 + 70 * u[:, 0] ** 2).astype(np.float32)       # forward (some behind the camera)
    velo[:, 1] = (-25 + 50 * u[:, 1]).astype(np.float32)
    velo[:, 2] = (-2.0 + 3.0 * u[:, 2]).astype(np.float32)
    velo[:, 3] = 1.0
    return p_rect, r_rect, r, t, velo


def gold_kitti(ref_kitti):
    p_rect, r_rect, r, t, velo = synthetic_kitti_scene()
    with tempfile.TemporaryDirectory() as tmp:
        tmp = pathlib.Path(tmp)
        fmt = lambda a: " ".join("%.6e" % v for v in a)
        (tmp / "calib_cam_to_cam.txt").write_text(
            "calib_time: 09-Jan-2012 13:57:47\nR_rect_00: %s\nP_rect_02: %s\n" % (fmt(r_rect), fmt(p_rect)))
        (tmp / "calib_velo_to_cam.txt").write_text(
            "calib_time: 15-Mar-2012 11:37:16\nR: %s\nT: %s\n" % (fmt(r), fmt(t)))
        velo.tofile(str(tmp / "scan.bin"))
        arrays = {}
        for shape in ((375, 1242), (120, 400)):
            depth = ref_kitti.generate_depth_map(tmp, tmp / "scan.bin", shape, cam=2)
            mask = ref_kitti.generate_mask(depth, 1e-3, 80)
            yy, xx = np.nonzero(depth)
            arrays["depth:%dx%d:yx" % shape] = np.stack([yy, xx], 1).astype(np.int32)
            arrays["depth:%dx%d:val" % shape] = depth[yy, xx]
            arrays["mask:%dx%d:count" % shape] = np.int64(mask.sum())
            arrays["mask:%dx%d:rowsum" % shape] = mask.sum(1).astype(np.int32)
        arrays["sub2ind"] = np.array([ref_kitti.sub2ind((375, 1242), 10.0, 7.0), ref_kitti.sub2ind((120, 400), 119.0, 399.0)])
    save("kitti_gt", **arrays)


def _grad_summaries(net, arrays, keys, prefix="grad:"):
    params = dict(net.named_parameters())
    for key in keys:
        g = params[key].grad
        if g is None:
            arrays[prefix + key + ":none"] = np.int64(1)
            continue
        for k, v in detgen.summarize(g, stride=53).items():
            arrays["%s%s:%s" % (prefix, key, k)] = v


def gold_res50(ref_res50, ref_loss):
    """BASELINE config 4 net: Disp_res_50 forward (4 outputs, full at a small shape), l1+smooth backward, bn1 quirk, eval."""
    b, h, w = 2, 64, 96
    net = ref_res50.Disp_res_50(datasets="nyu")
    detgen.fill_state_dict(net.state_dict(), "res50")
    x = detgen.image_batch(b, h, w, "res50:x")
    gt = detgen.sparse_depth(b, h, w, "res50:gt", density=0.6, lo=0.3, hi=11.0)
    net.train()
    disps = net(x)
    depth = [1 / d for d in disps]
    loss = ref_loss.l1_loss(gt, depth, "nyu") + 0.1 * ref_loss.smooth_loss(depth)
    loss.backward()
    arrays = {"loss": np.float64(loss.item())}
    for i, o in enumerate(disps):
        arrays["disp%d" % i] = _np(o)
    _grad_summaries(net, arrays, ["conv1.weight", "bn1.weight", "layer1.0.conv1.weight", "layer1.0.downsample.0.weight",
                                  "layer2.0.bn3.weight", "layer3.5.conv2.weight", "layer4.2.bn3.bias", "upconv5.0.weight",
                                  "iconv3.0.weight", "iconv1.0.bias", "predict_disp1.0.weight"])
    sd = net.state_dict()
    for key in ("bn1.running_mean", "bn1.running_var", "layer4.2.bn3.running_mean", "layer1.0.downsample.1.running_var"):
        arrays["bn:" + key] = _np(sd[key])
    arrays["bn1.num_batches_tracked"] = np.int64(int(sd["bn1.num_batches_tracked"]))
    net.eval()
    with torch.no_grad():
        arrays["eval_disp1"] = _np(net(x))
    save("res50", **arrays)


def gold_mono2(ref_networks, ref_mono2):
    """monodepth2-style nets: vggEncoder(16)+DepthDecoder and ResnetEncoder(18)+DepthDecoder through models.monodepth2."""
    b, h, w = 2, 64, 96
    x = (detgen.image_batch(b, h, w, "mono2:x") + 1) / 2          # monodepth2 consumes [0,1] images (train.py:122-127)
    arrays = {}
    for tag, enc in (("vgg", ref_networks.vggEncoder(16, False)), ("res18", ref_networks.ResnetEncoder(18, False))):
        dec = ref_networks.DepthDecoder(enc.num_ch_enc)
        net = ref_mono2.monodepth2(enc, dec)
        detgen.fill_state_dict(net.state_dict(), "mono2:" + tag)
        net.train()
        outs = net(x)
        ws = [detgen.uniform(tuple(o.shape), "mono2:g%d" % i, -1, 1) for i, o in enumerate(outs)]
        sum((o * wt).sum() for o, wt in zip(outs, ws)).backward()
        for i, o in enumerate(outs):
            arrays["%s:disp%d" % (tag, i)] = _np(o)
        keys = ["decoder.decoder.0.conv.conv.weight", "decoder.decoder.9.conv.conv.weight", "decoder.decoder.10.conv.weight",
                "decoder.decoder.13.conv.bias"]
        keys += ["encoder.encoder.features.0.weight", "encoder.encoder.features.41.weight"] if tag == "vgg" else \
                ["encoder.encoder.conv1.weight", "encoder.encoder.bn1.weight", "encoder.encoder.layer2.0.downsample.0.weight",
                 "encoder.encoder.layer4.1.bn2.bias"]
        _grad_summaries(net, arrays, keys, prefix=tag + ":grad:")
        arrays[tag + ":keys"] = np.array(sorted(net.state_dict().keys()))
        net.eval()
        with torch.no_grad():
            arrays[tag + ":eval_disp0"] = _np(net(x))
    save("mono2", **arrays)


def gold_posenet(ref_pose):
    """PoseExpNet (config 3's pose / explainability producer): masks + pose, train and eval, with and without the mask decoder."""
    b, h, w = 2, 128, 416
    tgt = detgen.image_batch(b, h, w, "pose:tgt")
    refs = [detgen.image_batch(b, h, w, "pose:ref%d" % i) for i in range(2)]
    arrays = {}
    for exp in (False, True):
        net = ref_pose.PoseExpNet(nb_ref_imgs=2, output_exp=exp)
        detgen.fill_state_dict(net.state_dict(), "posenet")
        net.train()
        masks, pose = net(tgt, refs)
        tag = "exp%d" % int(exp)
        arrays[tag + ":pose"] = _np(pose)
        loss = (pose * detgen.uniform(tuple(pose.shape), "pose:gp", -1, 1)).sum()
        if exp:
            for i, m in enumerate(masks):
                for k, v in detgen.summarize(m).items():
                    arrays["%s:mask%d:%s" % (tag, i, k)] = v
                loss = loss + (m * detgen.uniform(tuple(m.shape), "pose:gm%d" % i, -1, 1)).sum()
        loss.backward()
        keys = ["conv1.0.weight", "conv7.0.weight", "pose_pred.weight", "pose_pred.bias"] + (["upconv5.0.weight", "predict_mask1.weight"] if exp else [])
        _grad_summaries(net, arrays, keys, prefix=tag + ":grad:")
        net.eval()
        with torch.no_grad():
            m1, pe = net(tgt, refs)
        arrays[tag + ":eval_pose"] = _np(pe)
        if exp:
            arrays[tag + ":eval_mask1_samples"] = detgen.summarize(m1)["samples"]
    save("posenet", **arrays)


def gold_config3(ref_vgg, ref_pose, ref_loss, ref_warp):
    """BASELINE config 3 at its own resolution (2 x 128 x 416, seq-len 3): train.py:426-488 -- PoseExpNet -> Disp_vgg_BN ->
    1/disp -> photometric_reconstruction_loss + 0.1 * smooth_loss -> backward through BOTH nets (grid_sample at the container's
    default align_corners=False, which is what the imported reference yields today)."""
    b, h, w = 2, 128, 416
    ref_warp.pixel_coords = None
    tgt = detgen.image_batch(b, h, w, "cfg3:tgt")
    refs = [(tgt + 0.1 * detgen.uniform((b, 3, h, w), "cfg3:ref%d" % i, -1, 1)).clamp(-1, 1) for i in range(2)]
    k = torch.tensor([[241.67, 0, 204.17], [0, 246.28, 59.0], [0, 0, 1]], dtype=torch.float32).repeat(b, 1, 1)
    kinv = torch.inverse(k)
    disp_net = ref_vgg.Disp_vgg_BN(datasets="kitti")
    detgen.fill_state_dict(disp_net.state_dict(), "vggbn")
    pose_net = ref_pose.PoseExpNet(nb_ref_imgs=2, output_exp=False)
    detgen.fill_state_dict(pose_net.state_dict(), "posenet")
    disp_net.train(); pose_net.train()
    mask, pose = pose_net(tgt, refs)
    disps = disp_net(tgt)
    depth = [1 / d for d in disps]
    l1 = ref_loss.photometric_reconstruction_loss(tgt, refs, k, kinv, depth, mask, pose, "euler", "zeros")
    l3 = ref_loss.smooth_loss(depth)
    (1.0 * l1 + 0.1 * l3).backward()
    arrays = {"photo": np.float64(l1.item()), "smooth": np.float64(l3.item()), "pose": _np(pose), "kinv": _np(kinv)}
    for i, o in enumerate(disps):
        for kk, v in detgen.summarize(o).items():
            arrays["disp%d_%s" % (i, kk)] = v
    _grad_summaries(disp_net, arrays, ["features.features.0.weight", "features.features.40.weight", "upconv4.0.weight", "iconv2.0.weight",
                                       "iconv0.0.weight", "iconv0.0.bias", "disp0.0.weight", "disp3.0.weight"], prefix="disp:grad:")
    _grad_summaries(pose_net, arrays, ["conv1.0.weight", "conv7.0.weight", "pose_pred.weight", "pose_pred.bias"], prefix="pose:grad:")
    g = dict(pose_net.named_parameters())["pose_pred.bias"].grad
    arrays["full:pose:grad:pose_pred.bias"] = _np(g)
    g = dict(disp_net.named_parameters())["disp0.0.weight"].grad
    arrays["full:disp:grad:disp0.0.weight"] = _np(g)
    save("config3_cfg", **arrays)


class _FixedChannelMask(nn.Module):
    """Stand-in for nn.Dropout2d(0.5) with a given keep/scale pattern (RNG streams cannot be matched across implementations)."""

    def __init__(self, mask):
        super().__init__()
        self.mask = mask

    def forward(self, x):
        return x * self.mask if self.training else x


def gold_dorn80(ref_dorn, ref_utils, ref_loss):
    """BASELINE config 5 at its own size: Disp_vgg_BN_DORN with ordinal_c = 80 (train.py:35 default) on 2 x 128 x 416, training
    mode with an injected Dropout2d pattern, DORN_loss + backward (fp32; the reference has no mixed precision)."""
    b, h, w, K = 2, 128, 416, 80
    net = ref_dorn.Disp_vgg_BN_DORN(datasets="kitti", ordinal_c=K)
    detgen.fill_state_dict(net.state_dict(), "vggdorn80")
    mask = detgen.bernoulli((b, 16), "dorn80:drop", 0.5).float() * 2.0
    net.dropout = _FixedChannelMask(mask.view(b, 16, 1, 1))
    net.train()
    x = detgen.image_batch(b, h, w, "dorn80:x")
    gt = detgen.sparse_depth(b, h, w, "dorn80:gt", density=0.05)
    tgt = ref_utils.get_labels_sid(gt, ordinal_c=K, dataset="kitti")
    dec, ordc = net(x)
    loss = ref_loss.DORN_loss(gt, ordc, tgt, "kitti")
    loss.backward()
    arrays = {"loss": np.float64(loss.item()), "labels_sum": np.int64(int(tgt.long().sum())), "decode_sum": np.int64(int(dec.sum())),
              "decode_samples": _np(dec).reshape(-1)[::997].astype(np.int64)}
    for kk, v in detgen.summarize(ordc).items():
        arrays["ord_%s" % kk] = v
    params = dict(net.named_parameters())
    arrays["full:grad:conv_ord.weight"] = _np(params["conv_ord.weight"].grad)
    arrays["full:grad:conv_ord.bias"] = _np(params["conv_ord.bias"].grad)
    _grad_summaries(net, arrays, ["iconv0.0.weight", "upconv0.0.weight", "features.features.40.weight", "features.features.0.weight"])
    save("dorn80_cfg", **arrays)


ZOO = (
    # tag, reference file, class, constructor kwargs, parameters whose gradient summaries are kept, BatchNorm buffers kept
    ("res18", "models/Disp_res_18.py", "Disp_res_18", {"datasets": "nyu"},
     ["conv1.weight", "bn1.weight", "layer1.0.conv1.weight", "layer2.0.downsample.0.weight", "layer3.1.bn2.weight", "layer4.1.conv2.weight",
      "upconv5.0.weight", "iconv3.0.weight", "iconv1.0.bias", "predict_disp1.0.weight"],
     ["bn1.running_mean", "layer4.1.bn2.running_var"]),
    ("res6", "models/Disp_res.py", "Disp_res", {"datasets": "kitti"},
     ["conv1.weight", "layer1.0.conv1.weight", "layer2.0.bn3.weight", "layer3.5.conv2.weight", "layer4.2.bn3.bias", "upconv6.0.weight",
      "iconv6.0.weight", "upconv3.0.weight", "iconv3.0.weight", "iconv2.0.weight", "iconv1.0.bias", "predict_disp3.0.weight"],
     ["bn1.running_var", "layer4.2.bn3.running_mean"]),
    ("res101", "models/Disp_res_101.py", "Disp_res_101", {"datasets": "kitti"},
     ["conv1.weight", "layer3.22.conv2.weight", "layer3.11.bn1.weight", "upconv6.0.weight", "iconv3.0.weight", "predict_disp1.0.weight"],
     ["layer3.22.bn3.running_mean"]),
    ("vgg", "models/Disp_vgg.py", "Disp_vgg", {"alpha": 10, "beta": 0.01},
     ["conv1.0.weight", "conv1.2.bias", "conv3.4.weight", "conv5.4.weight", "upconv4.0.weight", "iconv2.0.weight", "iconv0.0.weight",
      "disp0.0.weight", "disp3.0.bias"], []),
    ("vggfeat", "models/Disp_vgg_feature.py", "Disp_vgg_feature", {"datasets": "nyu"},
     ["features.features.0.weight", "features.features.14.weight", "features.features.28.bias", "upconv3.0.weight", "iconv1.0.weight",
      "disp1.0.weight"], []),
)


def gold_zoo(ref_loss):
    """Disp_res_18, Disp_res, Disp_res_101, Disp_vgg, Disp_vgg_feature -- forward (4 outputs, full, 2 x 64 x 96),
    l1 + 0.1 smooth backward (gradient summaries), BatchNorm buffers, eval output, state_dict key list."""
    b, h, w = 2, 64, 96
    arrays = {}
    for tag, rel, cls, kwargs, gkeys, bnkeys in ZOO:
        mod = _load("ref_zoo_" + tag, rel)
        net = getattr(mod, cls)(**kwargs)
        detgen.fill_state_dict(net.state_dict(), "zoo:" + tag)
        ds = kwargs.get("datasets", "kitti")
        x = detgen.image_batch(b, h, w, "zoo:%s:x" % tag)
        gt = detgen.sparse_depth(b, h, w, "zoo:%s:gt" % tag, density=0.6, lo=0.3, hi=11.0)
        net.train()
        disps = net(x)
        depth = [1 / d for d in disps]
        loss = ref_loss.l1_loss(gt, depth, ds) + 0.1 * ref_loss.smooth_loss(depth)
        loss.backward()
        arrays[tag + ":loss"] = np.float64(loss.item())
        arrays[tag + ":keys"] = np.array(sorted(net.state_dict().keys()))
        for i, o in enumerate(disps):
            arrays["%s:disp%d" % (tag, i)] = _np(o)
        _grad_summaries(net, arrays, gkeys, prefix=tag + ":grad:")
        sd = net.state_dict()
        for key in bnkeys:
            arrays["%s:bn:%s" % (tag, key)] = _np(sd[key])
        net.eval()
        with torch.no_grad():
            arrays[tag + ":eval"] = _np(net(x))
    save("zoo", **arrays)


# FCRN / ASPP (SURVEY 8 f-4 tail).  tag, reference file, class, constructor kwargs, gradient keys, BatchNorm buffer keys
ZOO2 = (
    ("fcrn", "models/FCRN.py", "FCRN", {"datasets": "kitti"},
     ["conv1.weight", "bn1.weight", "layer2.0.downsample.0.weight", "layer4.2.conv3.weight", "conv2.weight", "bn2.bias", "up1.conv1_1.weight",
      "up1.conv1_2.weight", "up1.conv1_3.bias", "up1.conv1_4.weight", "up1.conv2_2.weight", "up1.bn1_1.weight", "up1.bn1_2.bias",
      "up2.conv3.weight", "up2.bn2.weight", "up3.conv2_3.weight", "up4.conv2_4.weight", "up4.conv1_1.bias", "conv3.weight", "conv3.bias"],
     ["bn1.running_mean", "bn2.running_var", "up1.bn1_1.running_mean", "up1.bn1_2.running_var", "up4.bn2.running_mean"]),
    ("res50_aspp", "models/res_aspp.py", "res50_aspp", {"datasets": "kitti"},
     ["Scale.conv1.weight", "Scale.layer1.0.conv1.weight", "Scale.layer2.0.conv1.weight", "Scale.layer2.0.downsample.0.weight",
      "Scale.layer3.0.conv2.weight", "Scale.layer3.0.downsample.0.weight", "Scale.layer3.5.conv2.weight", "Scale.layer4.0.conv2.weight",
      "Scale.layer4.2.conv3.weight", "Scale.layer5.conv2d_list.0.weight", "Scale.layer5.conv2d_list.1.bias", "Scale.layer5.conv2d_list.3.weight",
      "Scale.bn1.weight"],
     ["Scale.bn1.running_mean", "Scale.layer3.0.bn2.running_var", "Scale.layer4.2.bn3.running_mean"]),
    ("deeplab", "models/ASPP.py", "deeplab_depth", {},
     ["Scale.conv1.weight", "Scale.layer3.22.conv2.weight", "Scale.layer3.11.conv1.weight", "Scale.layer4.1.conv2.weight",
      "Scale.layer5.conv2d_list.2.weight", "Scale.layer5.conv2d_list.0.bias"],
     ["Scale.layer3.22.bn2.running_var"]),
)


def gold_zoo2(ref_loss):
    """FCRN, res50_aspp, deeplab_depth: forward (one output at the input size, 2 x 64 x 96), l1 + 0.1 smooth backward (gradient summaries),
    BatchNorm buffers, eval output, state_dict key list.  FCRN's Dropout2d draws from the RNG: its keep pattern is injected."""
    b, h, w = 2, 64, 96
    arrays = {}
    for tag, rel, cls, kwargs, gkeys, bnkeys in ZOO2:
        mod = _load("ref_zoo2_" + tag, rel)
        net = getattr(mod, cls)(**kwargs)
        detgen.fill_state_dict(net.state_dict(), "zoo2:" + tag)
        if tag == "fcrn":
            mask = detgen.bernoulli((b, 64), "zoo2:fcrn:drop", 0.5).float() * 2.0
            net.drop = _FixedChannelMask(mask.view(b, 64, 1, 1))
        x = detgen.image_batch(b, h, w, "zoo2:%s:x" % tag)
        gt = detgen.sparse_depth(b, h, w, "zoo2:%s:gt" % tag, density=0.6, lo=0.3, hi=11.0)
        net.train()
        disps = net(x)
        depth = [1 / d for d in disps]
        loss = ref_loss.l1_loss(gt, depth, "kitti") + 0.1 * ref_loss.smooth_loss(depth)
        loss.backward()
        arrays[tag + ":loss"] = np.float64(loss.item())
        arrays[tag + ":keys"] = np.array(sorted(net.state_dict().keys()))
        arrays[tag + ":trainable"] = np.array(sorted(k for k, q in net.named_parameters() if q.requires_grad))
        arrays[tag + ":disp0"] = _np(disps[0])
        _grad_summaries(net, arrays, gkeys, prefix=tag + ":grad:")
        sd = net.state_dict()
        for key in bnkeys:
            arrays["%s:bn:%s" % (tag, key)] = _np(sd[key])
        net.eval()
        with torch.no_grad():
            arrays[tag + ":eval"] = _np(net(x))
    save("zoo2", **arrays)


def _reference_slice(rel, first, last, must_contain):
    """Source lines first..last (1-based, inclusive) of a reference file, dedented, for exec: the worst-pixel selection lives INLINE
    in test_disp.py's main() (it is not a function that could be imported), so the generator runs the reference's own statements on
    its inputs.  Nothing of it is stored: only the resulting numbers."""
    import textwrap
    lines = (REF / rel).read_text().splitlines()[first - 1:last]
    text = textwrap.dedent("\n".join(lines))
    for m in must_contain:
        assert m in text, "reference %s:%d-%d no longer holds %r" % (rel, first, last, m)
    return text


def abs_rel_inputs(shape, variant=0):
    """gt: the synthetic KITTI depth map of gold_kitti (sparse, from the reference's own generate_depth_map, passed in); pred: a smooth
    closed-form depth map both sides can evaluate (two variants)."""
    h, w = shape
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    if variant == 0:
        return 4.0 + 60.0 * (1.0 - yy / h) ** 2 + 3.0 * np.sin(xx * 0.05) + 2.0 * np.cos(yy * 0.11 + xx * 0.013)
    return 25.0 + 20.0 * np.sin(yy * 0.031 + 0.4) * np.cos(xx * 0.0071) + 0.01 * xx


def gold_worst_pixels(ref_kitti):
    """test_disp.py:471-477 (compute_abs_rel_per_pixel) and :318-338 (crop, np.where, argpartition(-300)) executed from the
    reference's source on the synthetic scene's ground truth."""
    ns = {"np": np}
    exec(_reference_slice("test_disp.py", 471, 477, ["def compute_abs_rel_per_pixel", "abs_rel[valid_complement] = -1"]), ns)
    block = _reference_slice("test_disp.py", 318, 338, ["valid = current_abs_rel_per_pixel>0", "np.argpartition(index_result[:,2], -300)[-300:]",
                                                        "graph_index = index_result[max_100_error_index,:2].astype(np.int32)"])
    tint = _reference_slice("test_disp.py", 345, 350, ["annotate_input = np.copy(sample['tgt'])", "255.0/2.0"])
    p_rect, r_rect, r, t, velo = synthetic_kitti_scene()
    arrays = {}
    with tempfile.TemporaryDirectory() as tmp:
        tmp = pathlib.Path(tmp)
        fmt = lambda a: " ".join("%.6e" % v for v in a)
        (tmp / "calib_cam_to_cam.txt").write_text(
            "calib_time: 09-Jan-2012 13:57:47\nR_rect_00: %s\nP_rect_02: %s\n" % (fmt(r_rect), fmt(p_rect)))
        (tmp / "calib_velo_to_cam.txt").write_text(
            "calib_time: 15-Mar-2012 11:37:16\nR: %s\nT: %s\n" % (fmt(r), fmt(t)))
        velo.tofile(str(tmp / "scan.bin"))
        for variant in (0, 1):
            shape = (375, 1242)
            gt = ref_kitti.generate_depth_map(tmp, tmp / "scan.bin", shape, cam=2)
            pred = abs_rel_inputs(shape, variant)
            with np.errstate(divide="ignore", invalid="ignore"):
                m = ns["compute_abs_rel_per_pixel"](gt, pred, min_depth=1e-3, max_depth=80)
            tgt = (np.arange(shape[0] * shape[1] * 3, dtype=np.int64) * 7919 % 256).reshape(shape + (3,)).astype(np.float32)
            loc = {"np": np, "current_abs_rel_per_pixel": m, "sample": {"tgt": tgt}}
            exec(block, loc)
            exec(tint, loc)
            tag = "%dx%d:v%d" % (shape + (variant,))
            yy, xx = np.nonzero(m > 0)
            arrays["abs_rel:%s:yx" % tag] = np.stack([yy, xx], 1).astype(np.int32)
            arrays["abs_rel:%s:val" % tag] = m[yy, xx]
            arrays["abs_rel:%s:neg_count" % tag] = np.int64((m == -1).sum())
            arrays["worst:%s:graph_index" % tag] = loc["graph_index"]                 # numpy's own output order
            arrays["worst:%s:n_valid" % tag] = np.int64(loc["index_result"].shape[0])
            arrays["annotate:%s:sum" % tag] = loc["annotate_input"].astype(np.float64).sum(axis=(0, 1))
            ys, xs = loc["graph_index"][:8, 0], loc["graph_index"][:8, 1]
            arrays["annotate:%s:patch" % tag] = np.stack([loc["annotate_input"][y:y + 5, x:x + 5] for y, x in zip(ys, xs)])
    save("worst_pixels", **arrays)


def eval_chain_image(h, w):
    """A float32 H x W x 3 image in [0, 255] by formula (what test_framework_KITTI hands out as sample['tgt'], already at the network's
    input size so that the reference's removed scipy.misc.imresize is not needed)."""
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    img = np.stack([127.5 + 120.0 * np.sin(0.021 * xx + 0.3) * np.cos(0.047 * yy),
                    127.5 + 110.0 * np.cos(0.013 * xx - 0.029 * yy + 1.1),
                    255.0 * ((xx * 7 + yy * 13) % 97) / 96.0], axis=2)
    return img.clip(0, 255).astype(np.float32)


def gold_eval_chain(ref_vgg, ref_kitti):
    """The per-image evaluation chain of the reference's test_disp.py main() (it is inline there, :178-398): image -> transpose ->
    normalise -> Disp_vgg_BN.eval() -> 1 / disp -> scipy zoom to the ground-truth size -> clip -> Garg-crop mask -> scale factor ->
    compute_errors.  The reference's own statements are executed (slices of its source, nothing stored) on a deterministic sample: a
    closed-form image at the network's input size (the `imresize` branch at :193-194 is not taken), the synthetic KITTI scene's ground
    truth and mask from the reference's generate_depth_map / generate_mask, a detgen-filled reference Disp_vgg_BN.  torchvision's
    Normalize(mean 0.5, std 0.5) at :214 is (x - 0.5) / 0.5 -- the one line restated here (torchvision is not in the container)."""
    from scipy.ndimage import zoom
    ns = {"np": np}
    exec(_reference_slice("test_disp.py", 453, 469, ["def compute_errors(gt, pred):", "return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3"]), ns)
    pre = _reference_slice("test_disp.py", 192, 200, ["h,w,_ = tgt_img.shape", "tgt_img = np.transpose(tgt_img, (2, 0, 1))", "torch.from_numpy(tgt_img)"])
    fwd = _reference_slice("test_disp.py", 242, 242, ["pred_disp = disp_net(tgt_img).cpu().numpy()[0,0]"])
    post = _reference_slice("test_disp.py", 254, 266, ["gt_depth = sample['gt_depth']", "pred_depth = 1/pred_disp", "zoom(pred_depth,", ").clip(min_depth, max_depth)"])
    msk = _reference_slice("test_disp.py", 307, 308, ["pred_depth_zoomed = pred_depth_zoomed[sample['mask']]", "gt_depth = gt_depth[sample['mask']]"])
    fin = _reference_slice("test_disp.py", 391, 398, ["scale_factor = np.median(gt_depth)/np.median(pred_depth_zoomed)", "scale_factor = 5.4",
                                                      "errors[1,:,j] = compute_errors(gt_depth, pred_depth_zoomed*scale_factor)"])
    net = ref_vgg.Disp_vgg_BN(datasets="kitti")
    detgen.fill_state_dict(net.state_dict(), "vggbn")
    net.eval()
    p_rect, r_rect, r, t, velo = synthetic_kitti_scene()
    arrays = {}
    with tempfile.TemporaryDirectory() as tmp:
        tmp = pathlib.Path(tmp)
        fmt = lambda a: " ".join("%.6e" % v for v in a)
        (tmp / "calib_cam_to_cam.txt").write_text(
            "calib_time: 09-Jan-2012 13:57:47\nR_rect_00: %s\nP_rect_02: %s\n" % (fmt(r_rect), fmt(p_rect)))
        (tmp / "calib_velo_to_cam.txt").write_text(
            "calib_time: 15-Mar-2012 11:37:16\nR: %s\nT: %s\n" % (fmt(r), fmt(t)))
        velo.tofile(str(tmp / "scan.bin"))
        gt = ref_kitti.generate_depth_map(tmp, tmp / "scan.bin", (375, 1242), cam=2)
    mask = ref_kitti.generate_mask(gt, 1e-3, 80)
    for name, flags in (("supervised", {}), ("median", {"unsupervised": True}), ("stereo", {"stereo": True})):
        args = types.SimpleNamespace(no_resize=False, gt_type="KITTI", unsupervised=False, mono=False, stereo=False)
        for k, v in flags.items():
            setattr(args, k, v)
        loc = {"np": np, "torch": torch, "zoom": zoom, "args": args, "img_height": 128, "img_width": 416, "min_depth": 1e-3, "max_depth": 80,
               "sample": {"tgt": eval_chain_image(128, 416), "gt_depth": gt.copy(), "mask": mask}, "disp_net": net,
               "compute_errors": ns["compute_errors"], "errors": np.zeros((2, 7, 1), np.float32), "j": 0, "seq_length": 0}
        loc["tgt_img"] = loc["sample"]["tgt"]
        exec(pre, loc)
        loc["tgt_img"] = ((loc["tgt_img"] / 255 - 0.5) / 0.5).unsqueeze(0)          # test_disp.py:214,219 for KITTI
        with torch.no_grad():
            exec(fwd, loc)
        exec(post, loc)
        if name == "supervised":
            arrays["pred_depth_samples"] = loc["pred_depth"].reshape(-1)[::97].astype(np.float32)
            arrays["pred_depth_sum"] = np.float64(loc["pred_depth"].astype(np.float64).sum())
            z = loc["pred_depth_zoomed"]
            arrays["zoomed_shape"] = np.array(z.shape)
            arrays["zoomed_samples"] = z.reshape(-1)[::9973].astype(np.float32)
            arrays["zoomed_sum"] = np.float64(z.astype(np.float64).sum())
        exec(msk, loc)
        exec(fin, loc)
        arrays["errors:" + name] = loc["errors"][1, :, 0].astype(np.float64)
        arrays["scale:" + name] = np.float64(loc["scale_factor"])
        arrays["n_valid"] = np.int64(loc["gt_depth"].size)
    save("eval_chain", **arrays)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    _install_shims()
    ref_dispnets = _load("ref_dispnets", "models/DispNetS.py")
    ref_vgg = _load("ref_vgg_bn", "models/Disp_vgg_BN.py")
    ref_dorn = _load("ref_vgg_bn_dorn", "models/Disp_vgg_BN_DORN.py")
    ref_warp = _load("inverse_warp", "inverse_warp.py")      # loss_functions imports it by this name
    ref_loss = _load("ref_loss_functions", "loss_functions.py")
    ref_layers = _load("ref_layers", "layers.py")
    ref_utils = _load("ref_utils", "utils.py")
    ref_kitti = _load("ref_kitti_eval", "kitti_eval/depth_evaluation_utils.py")
    ref_res50 = _load("ref_disp_res_50", "models/Disp_res_50.py")
    ref_pose = _load("ref_poseexpnet", "models/PoseExpNet.py")
    ref_mono2 = _load("ref_monodepth2", "models/monodepth2.py")
    sys.modules["layers"] = ref_layers                         # networks/depth_decoder.py does `from layers import *`
    import networks as ref_networks                            # the reference package (sys.path[0] is the reference root)
    want = set(sys.argv[1:])
    sections = {
        "dispnets": lambda: gold_dispnets(ref_dispnets),
        "vgg": lambda: gold_vgg_bn(ref_vgg, ref_loss),
        "trainstep": lambda: gold_train_step(ref_vgg, ref_loss),
        "losses": lambda: gold_losses(ref_loss),
        "errors": lambda: gold_compute_errors(ref_loss),
        "warp": lambda: gold_warp(ref_warp, ref_loss),
        "layers": lambda: gold_layers(ref_layers),
        "dorn": lambda: gold_dorn(ref_dorn, ref_utils, ref_loss),
        "kitti": lambda: gold_kitti(ref_kitti),
        "res50": lambda: gold_res50(ref_res50, ref_loss),
        "mono2": lambda: gold_mono2(ref_networks, ref_mono2),
        "posenet": lambda: gold_posenet(ref_pose),
        "config3": lambda: gold_config3(ref_vgg, ref_pose, ref_loss, ref_warp),
        "dorn80": lambda: gold_dorn80(ref_dorn, ref_utils, ref_loss),
        "zoo": lambda: gold_zoo(ref_loss),
        "zoo2": lambda: gold_zoo2(ref_loss),
        "worst": lambda: gold_worst_pixels(ref_kitti),
        "evalchain": lambda: gold_eval_chain(ref_vgg, ref_kitti),
    }
    for name, fn in sections.items():
        if not want or name in want:
            print("==", name)
            fn()


if __name__ == "__main__":
    main()
