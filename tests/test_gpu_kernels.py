"""GPU parity of the individual HIP kernel families, called through the C ABI (ctypes) via the engine helpers,
against plain PyTorch-CPU fp32 references of the same op (F.conv2d & co).  Run on the MI355X box: pytest -m gpu.

Tolerances: the MFMA fp32 path is an exact fp32 FMA chain with a different summation order than ATen's CPU kernels,
so results agree to fp32 round-off: rtol 2e-4 / atol scaled to the magnitude of the result (stated per check).
"""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

from supervised_dispnet_amd import _lib, engine  # noqa: E402
from supervised_dispnet_amd._lib import ACT_ELU, ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_SIGMOID_AFFINE  # noqa: E402

DEV = torch.device("cuda:0")


def close(name, got, want, rtol=2e-4, atol_rel=2e-5):
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    assert got.shape == want.shape, "%s: shape %s vs %s" % (name, tuple(got.shape), tuple(want.shape))
    scale = float(want.abs().max()) + 1e-30
    err = (got - want).abs()
    tol = atol_rel * scale + rtol * want.abs()
    bad = err > tol
    if bad.any():
        idx = np.unravel_index(int(torch.argmax(err - tol)), got.shape)
        raise AssertionError("%s: %d/%d elements off; worst at %s got %.7g want %.7g (max|want| %.4g, max err %.4g)" % (
            name, int(bad.sum()), got.numel(), idx, float(got[idx]), float(want[idx]), scale, float(err.max())))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def rnd(*shape, lo=-1.0, hi=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


def test_library_loads_on_gfx950():
    lib = _lib.load()
    assert lib.dn_version() == _lib.EXPECTED_ABI
    assert lib.dn_device_arch_ok() == 1, "the HIP kernels are built for gfx950 only"


CONV_CASES = [
    # name,               cins,         ups,               cout, k, s, p, transposed, out_pad, N, H, W, act, affine
    ("3x3_64_64",         (64,),        (False,),          64, 3, 1, 1, False, 0, 2, 12, 20, ACT_NONE, False),
    ("3x3_first_nchw",    (3,),         (False,),          64, 3, 1, 1, False, 0, 2, 16, 24, ACT_NONE, False),
    ("3x3_first_tiles_nchw", (3,),      (False,),          64, 3, 1, 1, False, 0, 2, 24, 64, ACT_NONE, False),     # stem3 / lds3 wgrad: 8x32 tiles
    ("3x3_128_256_bnload", (128,),      (False,),          256, 3, 1, 1, False, 0, 2, 8, 12, ACT_NONE, True),
    ("3x3_wino_cat",      (32, 64),     (False, False),    96, 3, 1, 1, False, 0, 3, 18, 22, ACT_LEAKY, False),
    ("3x3_wino_cat_aff",  (64, 16),     (False, False),    64, 3, 1, 1, False, 0, 2, 20, 26, ACT_RELU, True),
    ("3x3_wino_512",      (512,),       (False,),          128, 3, 1, 1, False, 0, 6, 8, 26, ACT_NONE, False),
    ("3x3_wino_cat64_aff", (64, 128),   (False, False),    64, 3, 1, 1, False, 0, 2, 20, 30, ACT_LEAKY, True),
    ("3x3_wino_cat64",    (128, 64),    (False, False),    128, 3, 1, 1, False, 0, 3, 18, 22, ACT_NONE, False),
    ("3x3_wino_64_64",    (64,),        (False,),          64, 3, 1, 1, False, 0, 2, 24, 44, ACT_NONE, False),
    ("3x3_wino_t192",     (64,),        (False,),          64, 3, 1, 1, False, 0, 2, 16, 24, ACT_LEAKY, False),   # 192 tiles: the floor
    ("3x3_wino_t208_512", (512,),       (False,),          512, 3, 1, 1, False, 0, 4, 8, 26, ACT_NONE, True),     # conv5_x of a 4-image shard
    ("3x3_wino_t208_cat", (256, 512),   (False, False),    256, 3, 1, 1, False, 0, 4, 8, 26, ACT_LEAKY, False),   # iconv4 of a 4-image shard
    ("3x3_wino_cat_193",  (64, 128, 1), (False, False, True), 64, 3, 1, 1, False, 0, 2, 24, 44, ACT_LEAKY, False),
    ("3x3_wino_cat_97",   (32, 64, 1),  (False, False, True), 32, 3, 1, 1, False, 0, 2, 24, 44, ACT_LEAKY, False),
    ("3x3_wino_iconv2",   (64, 128, 1), (False, False, True), 64, 3, 1, 1, False, 0, 2, 32, 104, ACT_LEAKY, False),
    ("3x3_wino_iconv1",   (32, 64, 1),  (False, False, True), 32, 3, 1, 1, False, 0, 2, 64, 208, ACT_LEAKY, False),
    ("3x3_cat_193",       (64, 128, 1), (False, False, True), 64, 3, 1, 1, False, 0, 2, 8, 12, ACT_LEAKY, False),
    ("3x3_cat_17",        (16, 1),      (False, True),     16, 3, 1, 1, False, 0, 1, 16, 24, ACT_LEAKY, False),
    ("3x3_cat_17_tiles",  (16, 1),      (False, True),     16, 3, 1, 1, False, 0, 3, 20, 70, ACT_LEAKY, False),   # lds3: several ragged 8x32 tiles
    ("3x3_16_16_tiles",   (16,),        (False,),          16, 3, 1, 1, False, 0, 2, 24, 40, ACT_RELU, False),
    ("3x3_head",          (32,),        (False,),          1, 3, 1, 1, False, 0, 2, 10, 14, ACT_SIGMOID_AFFINE, False),
    ("3x3_head_16",       (16,),        (False,),          1, 3, 1, 1, False, 0, 3, 17, 23, ACT_SIGMOID_AFFINE, False),
    ("3x3_head_128",      (128,),       (False,),          1, 3, 1, 1, False, 0, 2, 5, 9, ACT_SIGMOID_AFFINE, False),
    ("3x3_head_64_tiles", (64,),        (False,),          1, 3, 1, 1, False, 0, 3, 21, 70, ACT_SIGMOID_AFFINE, False),    # 4 x 16 tiles, four channel groups per pixel, ragged
    ("3x3_head_16_big",   (16,),        (False,),          1, 3, 1, 1, False, 0, 4, 125, 1000, ACT_SIGMOID_AFFINE, False), # 8 x 64 tiles (>= 1024 blocks)
    ("3x3_head_32_big",   (32,),        (False,),          1, 3, 1, 1, False, 0, 4, 128, 1010, ACT_SIGMOID_AFFINE, False), # 8 x 64 tiles, two channel groups
    ("3x3_8_8",           (8,),         (False,),          24, 3, 1, 1, False, 0, 2, 9, 11, ACT_LEAKY, False),
    ("7x7_s2",            (3,),         (False,),          32, 7, 2, 3, False, 0, 2, 20, 28, ACT_RELU, False),
    ("5x5_s2",            (32,),        (False,),          64, 5, 2, 2, False, 0, 2, 18, 22, ACT_RELU, False),
    ("3x3_s2_odd",        (64,),        (False,),          128, 3, 2, 1, False, 0, 2, 13, 7, ACT_RELU, False),
    ("1x1",               (64,),        (False,),          160, 1, 1, 0, False, 0, 2, 6, 10, ACT_NONE, False),
    # 128 x 128 three-piece tile (round 4).  Smooth activations at this size: with 6 M outputs a ReLU whose pre-activation lies within fp32
    # round-off of zero flips against the CPU reference somewhere, and ONE flipped (pixel, channel) moves 256 entries of dw by ~0.2
    ("1x1_256_256_x3b",   (256,),       (False,),          256, 1, 1, 0, False, 0, 8, 48, 64, ACT_ELU, False),
    ("1x1_128_512_x3b_aff", (128,),     (False,),          512, 1, 1, 0, False, 0, 4, 48, 64, ACT_NONE, True),
    ("3x3_s2_128_256_x3b", (128,),      (False,),          256, 3, 2, 1, False, 0, 8, 96, 128, ACT_NONE, False),
    ("convT_k4s2p1_128_128_x3b", (128,), (False,),         128, 4, 2, 1, True, 0, 4, 32, 48, ACT_LEAKY, False),
    ("3x3_elu",           (16,),        (False,),          16, 3, 1, 1, False, 0, 1, 8, 8, ACT_ELU, False),
    ("convT_k4s2p1",      (64,),        (False,),          32, 4, 2, 1, True, 0, 2, 6, 10, ACT_LEAKY, False),
    ("convT_k4s2p1_big",  (512,),       (False,),          256, 4, 2, 1, True, 0, 2, 4, 13, ACT_LEAKY, False),
    ("convT_k4s2p1_64_32_tiles", (64,), (False,),          32, 4, 2, 1, True, 0, 8, 30, 100, ACT_LEAKY, False),   # lds3k: upconv1 (8-wave roles, ragged tiles)
    ("3x3_iconv1_tiles",  (32, 64, 1),  (False, False, True), 32, 3, 1, 1, False, 0, 3, 60, 200, ACT_LEAKY, False),  # lds3k: K-split roles, ragged tiles
    ("convT_k4s2p1_32_16", (32,),       (False,),          16, 4, 2, 1, True, 0, 2, 9, 21, ACT_LEAKY, False),     # lds3: upconv0 (four phases / stride-2 gather)
    ("convT_k4s2p1_32_16_tiles", (32,), (False,),          16, 4, 2, 1, True, 0, 3, 20, 70, ACT_LEAKY, False),
    ("convT_k3s2p1op1",   (32,),        (False,),          16, 3, 2, 1, True, 1, 2, 5, 7, ACT_RELU, False),
    ("7x7_s2_res_first_nchw", (3,),     (False,),          64, 7, 2, 3, False, 0, 4, 128, 512, ACT_NONE, False),   # stemk: ResNet conv1 form (16 x 32 tiles)
    ("7x7_s2_pose_first_nchw", (3, 3, 3), (False, False, False), 16, 7, 2, 3, False, 0, 2, 128, 512, ACT_LEAKY, False),   # stemk: PoseExpNet conv1 form, three NCHW operands
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_family_fwd_bwd(case):
    name, cins, ups, cout, k, s, p, transposed, out_pad, N, H, W, act, affine = case
    torch.manual_seed(1)
    mod = (nn.ConvTranspose2d(sum(cins), cout, k, s, p, out_pad) if transposed else nn.Conv2d(sum(cins), cout, k, s, p))
    with torch.no_grad():
        mod.bias.uniform_(-0.5, 0.5)
    # ---- reference (CPU fp32)
    srcs, ref_in = [], []
    for i, (c, up) in enumerate(zip(cins, ups)):
        h, w = (H // 2, W // 2) if up else (H, W)
        t = rnd(N, c, h, w, seed=i).requires_grad_()
        srcs.append(t)
        v = t
        if affine:
            sc, sh = rnd(c, lo=0.5, hi=1.5, seed=10 + i), rnd(c, lo=-0.3, hi=0.3, seed=20 + i)
            v = F.relu(v * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
        if up:
            v = F.interpolate(v, scale_factor=2, mode="nearest")
        ref_in.append(v)
    y_ref_pre = mod(torch.cat(ref_in, 1))
    crop = None
    if transposed and out_pad:
        crop = (2 * H - 1, 2 * W)   # exercise crop_like on one axis
        y_ref_pre = y_ref_pre[:, :, :crop[0], :crop[1]]
    p0, p1 = (0.1, 0.0) if act == ACT_LEAKY else ((10.0, 0.01) if act == ACT_SIGMOID_AFFINE else (0.0, 0.0))
    y_ref = {ACT_NONE: lambda v: v, ACT_RELU: F.relu, ACT_LEAKY: lambda v: F.leaky_relu(v, 0.1), ACT_ELU: F.elu,
             ACT_SIGMOID_AFFINE: lambda v: 10.0 * torch.sigmoid(v) + 0.01}[act](y_ref_pre)
    g = rnd(*y_ref.shape, seed=99)
    (y_ref * g).sum().backward()
    # ---- HIP
    mod_d = (nn.ConvTranspose2d(sum(cins), cout, k, s, p, out_pad) if transposed else nn.Conv2d(sum(cins), cout, k, s, p)).to(DEV)
    mod_d.load_state_dict(mod.state_dict())
    layer = engine.ConvLayer(mod_d[0] if isinstance(mod_d, nn.Sequential) else mod_d, transposed=transposed)
    pieces = []
    for i, (t, up) in enumerate(zip(srcs, ups)):
        n_, c_, h_, w_ = t.shape
        if name.endswith("nchw"):
            a = engine.Act.from_nchw_image(t.detach().to(DEV))
            a.needs_grad = False
        else:
            a = engine.Act(nhwc(t.detach()), n_, h_, w_, c_)
        if affine:
            a.scale = rnd(c_, lo=0.5, hi=1.5, seed=10 + i).to(DEV)
            a.shift = rnd(c_, lo=-0.3, hi=0.3, seed=20 + i).to(DEV)
        pieces.append(engine.Piece(a, up))
    tape, sink = engine.Tape(True), engine.GradSink()
    y = engine.block_conv_act(tape, sink, pieces, layer, act, p0, p1, out_hw=crop)
    torch.cuda.synchronize()
    close(name + ":y", nchw(y.t), y_ref)
    y.grad = nhwc(g)
    tape.run_backward()
    torch.cuda.synchronize()
    close(name + ":dw", sink.get(mod_d.weight), mod.weight.grad, rtol=5e-4, atol_rel=5e-5)
    close(name + ":db", sink.get(mod_d.bias), mod.bias.grad, rtol=5e-4, atol_rel=5e-5)
    for i, (pc, t) in enumerate(zip(pieces, srcs)):
        if not pc.act.needs_grad:
            continue
        if affine:
            continue   # gradient w.r.t. the pre-affine tensor goes through the BN path, tested in test_conv_bn_pool_block
        close(name + ":dx%d" % i, nchw(pc.act.grad), t.grad, rtol=5e-4, atol_rel=5e-5)


@pytest.mark.parametrize("shape", [(2, 8, 12, 3, 64, 64), (2, 16, 8, 64, 128, 128), (3, 18, 22, 16, 64, 96), (2, 24, 44, 64, 64, 128)],
                         ids=["c3_64_64", "c64_128_128", "c16_64_96_wino_ragged", "c64_64_128_wino_wgrad"])
def test_conv_bn_pool_block(shape):
    """conv -> BN(train) -> ReLU -> conv -> BN -> ReLU -> MaxPool, forward + full backward vs torch modules (CPU)."""
    N, H, W, c0, c1, c2 = shape
    torch.manual_seed(3)
    ref = nn.Sequential(nn.Conv2d(c0, c1, 3, padding=1), nn.BatchNorm2d(c1), nn.ReLU(), nn.Conv2d(c1, c2, 3, padding=1),
                        nn.BatchNorm2d(c2), nn.ReLU(), nn.MaxPool2d(2, 2))
    with torch.no_grad():
        for m in ref:
            if isinstance(m, nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.2)
    dev_mods = nn.Sequential(nn.Conv2d(c0, c1, 3, padding=1), nn.BatchNorm2d(c1), nn.ReLU(), nn.Conv2d(c1, c2, 3, padding=1),
                             nn.BatchNorm2d(c2), nn.ReLU(), nn.MaxPool2d(2, 2)).to(DEV)
    dev_mods.load_state_dict(ref.state_dict())
    x = rnd(N, c0, H, W, seed=5).requires_grad_()
    ref.train()
    out_ref = ref(x)
    g = rnd(*out_ref.shape, seed=6)
    (out_ref * g).sum().backward()
    tape, sink = engine.Tape(True), engine.GradSink()
    xa = engine.Act(nhwc(x.detach()), N, H, W, c0)
    y1 = engine.block_conv_bn(tape, sink, engine.Piece(xa), engine.ConvLayer(dev_mods[0]), dev_mods[1], True)
    y2 = engine.block_conv_bn(tape, sink, engine.Piece(y1), engine.ConvLayer(dev_mods[3]), dev_mods[4], True)
    pooled = engine.block_pool(tape, y2)
    torch.cuda.synchronize()
    close("pooled", nchw(pooled.t), out_ref)
    close("running_mean", dev_mods[4].running_mean, ref[4].running_mean, rtol=1e-4)
    close("running_var", dev_mods[4].running_var, ref[4].running_var, rtol=1e-4)
    assert int(dev_mods[1].num_batches_tracked) == 1
    pooled.grad = nhwc(g)
    tape.run_backward()
    torch.cuda.synchronize()
    for i in (0, 3):
        close("dw%d" % i, sink.get(dev_mods[i].weight), ref[i].weight.grad, rtol=1e-3, atol_rel=1e-4)
    for i in (1, 4):
        close("dgamma%d" % i, sink.get(dev_mods[i].weight), ref[i].weight.grad, rtol=1e-3, atol_rel=1e-4)
        close("dbeta%d" % i, sink.get(dev_mods[i].bias), ref[i].bias.grad, rtol=1e-3, atol_rel=1e-4)
    close("dx", nchw(xa.grad), x.grad, rtol=1e-3, atol_rel=1e-4)
    # conv bias in front of BN: exact zeros here, rounding noise in the reference
    assert float(sink.get(dev_mods[0].bias).abs().max()) == 0.0
    assert float(ref[0].bias.grad.abs().max()) < 1e-3      # pure rounding noise; its size depends on ATen's summation order


@pytest.mark.parametrize("form", ["res", "pose"])
def test_stemk_first_layers(form):
    """The 7x7 / stride-2 first layers on NCHW images (dn::stemk_conv_kernel): ResNet conv1 (3 -> 64) with the BatchNorm partial
    statistics of its epilogue -- BatchNorm(train) + ReLU of the result and the running statistics against torch --, and PoseExpNet
    conv1 (three 3-channel images -> 16, ReLU).  The kernel that ran is named."""
    torch.manual_seed(11)
    N, H, W, cins, cout = (4, 128, 512, (3,), 64) if form == "res" else (2, 128, 512, (3, 3, 3), 16)
    mod = nn.Conv2d(sum(cins), cout, 7, 2, 3)
    xs = [rnd(N, c, H, W, seed=30 + i) for i, c in enumerate(cins)]
    mod_d = nn.Conv2d(sum(cins), cout, 7, 2, 3).to(DEV)
    mod_d.load_state_dict(mod.state_dict())
    layer = engine.ConvLayer(mod_d)
    pieces = [engine.Piece(engine.Act.from_nchw_image(x.to(DEV))) for x in xs]
    act = ACT_NONE if form == "res" else ACT_RELU
    y_t, _, _ = engine.conv_forward(layer, pieces, act)
    assert "stemk_conv_kernel" in _lib.load().dn_last_kernel().decode()
    pre = mod(torch.cat(xs, 1))
    close(form + ":y", nchw(y_t), pre if form == "res" else F.relu(pre))
    # weight gradient of the same layer: dn::stemk_wgrad_kernel + the fixed-order slab fold
    g = rnd(*pre.shape, seed=40)
    (pre * g).sum().backward()
    dw = engine.conv_wgrad(layer, pieces, nhwc(g), (H // 2, W // 2))
    assert "stemk_wgrad_kernel" in _lib.load().dn_last_kernel().decode()
    close(form + ":dw", dw, mod.weight.grad, rtol=5e-4, atol_rel=5e-5)
    if form == "res":
        bn = nn.BatchNorm2d(cout)
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.2, 0.2)
        bn_d = nn.BatchNorm2d(cout).to(DEV)
        bn_d.load_state_dict(bn.state_dict())
        bn.train()
        ref = F.relu(bn(pre))
        tape, sink = engine.Tape(True), engine.GradSink()
        y = engine.block_conv_bn(tape, sink, pieces[0], layer, bn_d, True)
        torch.cuda.synchronize()
        out = torch.relu(y.t * y.scale + y.shift)
        close("res:bn_relu", nchw(out), ref, rtol=1e-3, atol_rel=1e-4)
        close("res:running_mean", bn_d.running_mean, bn.running_mean, rtol=1e-4)
        close("res:running_var", bn_d.running_var, bn.running_var, rtol=1e-4)


def test_winograd_path_is_taken_and_matches_direct(monkeypatch):
    """3x3/s1/p1 layers with 16-aligned channels run dn::wino_conv_kernel (forward and input gradient); the same call with
    DN_NO_WINOGRAD=1 runs the direct implicit GEMM: both agree to fp32 round-off (F(2x2,3x3) has the same error level)."""
    torch.manual_seed(4)
    N, H, W, cin, cout = 3, 18, 22, 64, 128
    mod = nn.Conv2d(cin, cout, 3, 1, 1).to(DEV)
    x = torch.randn(N, H, W, cin, device=DEV)
    dy = torch.randn(N, H, W, cout, device=DEV)
    res = {}
    for tag, env in (("wino", None), ("direct", "1")):
        if env is None:
            monkeypatch.delenv("DN_NO_WINOGRAD", raising=False)
        else:
            monkeypatch.setenv("DN_NO_WINOGRAD", env)
        _lib.load().dn_reload_knobs()
        engine.bump_param_epoch()
        layer = engine.ConvLayer(mod)
        xa = engine.Act(x.clone(), N, H, W, cin)
        y, _, _ = engine.conv_forward(layer, [engine.Piece(xa)], ACT_LEAKY, 0.1, 0.0)
        kf = _lib.load().dn_last_kernel().decode()
        engine.conv_dgrad(layer, dy, N, H, W, [engine.Piece(xa)], (H, W))
        kd = _lib.load().dn_last_kernel().decode()
        dw = engine.conv_wgrad(layer, [engine.Piece(xa)], dy, (H, W))
        kw = _lib.load().dn_last_kernel().decode()
        torch.cuda.synchronize()
        res[tag] = (y, xa.grad, kf, kd, dw, kw)
    assert "wino_conv" in res["wino"][2] and "wino_conv" in res["wino"][3] and "wino_wgrad" in res["wino"][5]
    assert "igemm" in res["direct"][2] and "igemm" in res["direct"][3] and "igemm" in res["direct"][5]
    close("wino_vs_direct:y", res["wino"][0], res["direct"][0], rtol=1e-4, atol_rel=1e-5)
    close("wino_vs_direct:dx", res["wino"][1], res["direct"][1], rtol=1e-4, atol_rel=1e-5)
    close("wino_vs_direct:dw", res["wino"][4], res["direct"][4], rtol=1e-4, atol_rel=1e-5)


@pytest.mark.parametrize("cfg", [(3, 18, 22, (128,), 128, False), (6, 8, 26, (256, 512), 256, True), (2, 20, 30, (64, 128), 128, True)],
                         ids=["128_128", "cat768_256_aff", "cat192_128_aff"])
def test_wide_winograd_wgrad_block_is_bitwise_the_narrow_one(monkeypatch, cfg):
    """dn::wino_wgrad_x3w_kernel (128 output x 64 input channels x 8 positions per block, round 4) against dn::wino_wgrad_x3_kernel
    (64 x 64 x 16 positions, DN_WINO_WGW=0): another partition of the same work -- every (position, co, ci) sum runs over the same
    tiles in the same chunk order with the same three-piece arithmetic -- so the two agree BIT FOR BIT, borders, ragged last chunk,
    virtual concat and pending BatchNorm + ReLU included."""
    N, H, W, cins, cout, aff = cfg
    torch.manual_seed(11)
    cin = sum(cins)
    mod = nn.Conv2d(cin, cout, 3, 1, 1).to(DEV)
    xs = [torch.randn(N, H, W, c, device=DEV) for c in cins]
    dy = torch.randn(N, H, W, cout, device=DEV)
    res = {}
    for tag, env in (("wide", None), ("narrow", "0")):
        if env is None:
            monkeypatch.delenv("DN_WINO_WGW", raising=False)
        else:
            monkeypatch.setenv("DN_WINO_WGW", env)
        _lib.load().dn_reload_knobs()
        engine.bump_param_epoch()
        layer = engine.ConvLayer(mod)
        pieces = []
        for i, (x, c) in enumerate(zip(xs, cins)):
            a = engine.Act(x.clone(), N, H, W, c)
            if aff and i == 0:
                g = torch.Generator(device="cpu").manual_seed(5)
                a.scale = (torch.rand(c, generator=g) + 0.5).to(DEV)
                a.shift = (torch.rand(c, generator=g) - 0.5).to(DEV)
            pieces.append(engine.Piece(a))
        dw = engine.conv_wgrad(layer, pieces, dy, (H, W))
        name = _lib.load().dn_last_kernel().decode()
        torch.cuda.synchronize()
        res[tag] = (dw.clone(), name)
    monkeypatch.delenv("DN_WINO_WGW", raising=False)
    _lib.load().dn_reload_knobs()
    assert "wino_wgrad_x3w_kernel" in res["wide"][1] and "wino_wgrad_x3_kernel" in res["narrow"][1], (res["wide"][1], res["narrow"][1])
    assert torch.equal(res["wide"][0], res["narrow"][0]), float((res["wide"][0] - res["narrow"][0]).abs().max())


@pytest.mark.parametrize("kind", ["iconv0", "upconv0", "iconv0_dgrad", "upconv0_dgrad"])
def test_thin_conv_matches_tiled_kernel(monkeypatch, kind):
    """The thin full-resolution layers at the metric's resolution on their three implementations: dn::lds3_conv_kernel (round 4: input
    tile resident in LDS as three bf16 pieces, DN_COMPUTE_F32X3 arithmetic), dn::thin_conv_kernel (DN_NO_LDS3=1: 16x16x4 fp32 MFMAs fed
    from global memory) and the tiled implicit GEMM (+ DN_NO_THIN_CONV=1).  Same fp32-level products, different summation order ->
    agreement to round-off everywhere, including the borders, the ragged last tile column and the 1-channel upsampled piece."""
    torch.manual_seed(5)
    N, H, W = 4, 128, 416
    res = {}
    for tag, envs in (("lds3", ()), ("thin", ("DN_NO_LDS3",)), ("tiled", ("DN_NO_LDS3", "DN_NO_THIN_CONV"))):
        for e in ("DN_NO_LDS3", "DN_NO_THIN_CONV"):
            monkeypatch.delenv(e, raising=False)
        for e in envs:
            monkeypatch.setenv(e, "1")
        _lib.load().dn_reload_knobs()
        torch.manual_seed(6)
        if kind.startswith("upconv0"):
            mod = nn.ConvTranspose2d(32, 16, 4, 2, 1).to(DEV)
            layer = engine.ConvLayer(mod, transposed=True)
            x = engine.Act(torch.randn(N, H // 2, W // 2, 32, device=DEV), N, H // 2, W // 2, 32)
            if kind == "upconv0":
                y, _, _ = engine.conv_forward(layer, [engine.Piece(x)], ACT_LEAKY, 0.1, 0.0)
                outs = [y]
            else:
                dy = torch.randn(N, H, W, 16, device=DEV)
                engine.conv_dgrad(layer, dy, N, H, W, [engine.Piece(x)], (H // 2, W // 2))
                outs = [x.grad]
        else:
            mod = nn.Conv2d(17, 16, 3, 1, 1).to(DEV)
            layer = engine.ConvLayer(mod)
            a = engine.Act(torch.randn(N, H, W, 16, device=DEV), N, H, W, 16)
            d = engine.Act(torch.rand(N, H // 2, W // 2, 1, device=DEV) * 10 + 0.01, N, H // 2, W // 2, 1)
            pieces = [engine.Piece(a), engine.Piece(d, up=True)]
            if kind == "iconv0":
                y, _, _ = engine.conv_forward(layer, pieces, ACT_LEAKY, 0.1, 0.0)
                outs = [y]
            else:
                dy = torch.randn(N, H, W, 16, device=DEV)
                engine.conv_dgrad(layer, dy, N, H, W, pieces, (H, W))
                outs = [a.grad, d.grad]
        name = _lib.load().dn_last_kernel().decode()
        torch.cuda.synchronize()
        res[tag] = (outs, name)
    for e in ("DN_NO_LDS3", "DN_NO_THIN_CONV"):
        monkeypatch.delenv(e, raising=False)
    _lib.load().dn_reload_knobs()
    assert "lds3_conv_kernel" in res["lds3"][1], res["lds3"][1]
    assert "thin_conv_kernel" in res["thin"][1] or kind.endswith("_dgrad")
    assert "thin" not in res["tiled"][1] and "lds3" not in res["tiled"][1]
    for tag in ("lds3", "thin"):
        for i, (got, want) in enumerate(zip(res[tag][0], res["tiled"][0])):
            close("%s:%s[%d]" % (tag, kind, i), got, want, rtol=2e-5, atol_rel=2e-6)


def test_winograd_error_vs_fp64(monkeypatch):
    """Rounding of F(2x2,3x3) against the direct fp32 FMA chain, both measured against an fp64 convolution of the same
    fp32 inputs: Winograd's transforms add a few ulps of the INPUT magnitude; stated bound: max error <= 4x the direct
    kernel's + 2e-6 of the result's magnitude (measured ~1.5-2.5x)."""
    torch.manual_seed(7)
    N, H, W, cin, cout = 2, 32, 48, 128, 64
    mod = nn.Conv2d(cin, cout, 3, 1, 1).to(DEV)
    x = torch.rand(N, H, W, cin, device=DEV)               # non-negative like a post-ReLU activation (the unfavourable case)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), mod.weight.detach().double().cpu(), mod.bias.detach().double().cpu(), padding=1)
    ref = ref.permute(0, 2, 3, 1)
    errs, l2 = {}, {}
    prev = engine.compute_mode()
    try:
        for tag, env, mode in (("wino", None, "f32"), ("direct", "1", "f32"), ("wino_f32x3", None, "f32x3")):
            if env is None:
                monkeypatch.delenv("DN_NO_WINOGRAD", raising=False)
            else:
                monkeypatch.setenv("DN_NO_WINOGRAD", env)
            _lib.load().dn_reload_knobs()
            engine.set_compute(mode)
            layer = engine.ConvLayer(mod)
            y, _, _ = engine.conv_forward(layer, [engine.Piece(engine.Act(x, N, H, W, cin))])
            torch.cuda.synchronize()
            assert ("wino_conv" in _lib.load().dn_last_kernel().decode()) == (env is None)
            errs[tag] = float((y.double().cpu() - ref).abs().max())
            l2[tag] = float((y.double().cpu() - ref).norm() / ref.norm())
    finally:
        engine.set_compute(prev)
    scale = float(ref.abs().max())
    print("max |err| vs fp64: winograd %.3g, direct %.3g, winograd on three-piece bf16 products %.3g (result magnitude %.3g)" % (
        errs["wino"], errs["direct"], errs["wino_f32x3"], scale))
    print("relative L2 vs fp64: winograd %.3g, direct %.3g, three-piece %.3g" % (l2["wino"], l2["direct"], l2["wino_f32x3"]))
    assert errs["wino"] <= 4.0 * errs["direct"] + 2e-6 * scale
    # the three-piece products are fp32 products: the same error level as the fp32 matrix instruction (stated: <= 1.5x + 2e-7 of the magnitude)
    assert errs["wino_f32x3"] <= 1.5 * errs["wino"] + 2e-7 * scale
    assert l2["wino_f32x3"] <= 1.5 * l2["wino"]


def _rel_l2(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return float((got - want).norm() / (want.norm() + 1e-300))


@pytest.mark.parametrize("mode", ["bf16", "f32x3"])
@pytest.mark.parametrize("case", ["plain_128_64", "bn_pending_64_128", "concat_64_64_1", "long_k_512_64"])
def test_winograd_compute_modes(case, mode):
    """dn_conv_desc.compute on the Winograd forward / input gradient, against the fp32 matrix-instruction kernel on the same inputs.
    bf16  (BASELINE configs[4]'s mixed precision): transformed tiles and weights ROUNDED to bf16, fp32 accumulation on
          v_mfma_f32_32x32x16_bf16.  Stated tolerance: relative L2 error <= 8e-3, max error <= 3 % of the result's magnitude
          (two operands rounded to 8 significant bits, 2^-9 rms each; measured 4e-3).
    f32x3 every fp32 operand split exactly into three bf16 pieces, six partial products, fp32 accumulation: an fp32 result.  Stated
          tolerance: the agreement two fp32 summation orders have (the same bound test_winograd_path_is_taken_and_matches_direct puts
          between the Winograd and the direct kernel): rtol 1e-4 / atol 1e-5 of the magnitude, relative L2 <= 2e-6.
    The kernel name proves which variant ran.  The weight gradient has an f32x3 variant too (dn::wino_wgrad_x3_kernel, same bound) and
    no bf16-rounded one (bit-identical to the fp32 mode there)."""
    torch.manual_seed(11)
    N, H, W = 3, 18, 22
    suffix = {"bf16": ", 1>", "f32x3": ", 3>"}[mode]
    prev = engine.compute_mode()
    try:
        res = {}
        for m in ("f32", mode):
            engine.set_compute(m)
            torch.manual_seed(12)
            if case == "plain_128_64":
                mod = nn.Conv2d(128, 64, 3, 1, 1).to(DEV)
                xa = engine.Act(torch.randn(N, H, W, 128, device=DEV), N, H, W, 128)
                pieces = [engine.Piece(xa)]
                cout = 64
            elif case == "long_k_512_64":
                mod = nn.Conv2d(512, 64, 3, 1, 1).to(DEV)
                xa = engine.Act(torch.rand(N, H, W, 512, device=DEV), N, H, W, 512)       # non-negative: no cancellation in the sums
                pieces = [engine.Piece(xa)]
                cout = 64
            elif case == "bn_pending_64_128":
                mod = nn.Conv2d(64, 128, 3, 1, 1).to(DEV)
                xa = engine.Act(torch.randn(N, H, W, 64, device=DEV), N, H, W, 64)
                xa.scale = torch.rand(64, device=DEV) + 0.5
                xa.shift = torch.randn(64, device=DEV) * 0.3
                pieces = [engine.Piece(xa)]
                cout = 128
            else:
                mod = nn.Conv2d(129, 64, 3, 1, 1).to(DEV)
                a = engine.Act(torch.randn(N, H, W, 64, device=DEV), N, H, W, 64)
                b = engine.Act(torch.randn(N, H, W, 64, device=DEV), N, H, W, 64)
                d = engine.Act(torch.rand(N, H // 2, W // 2, 1, device=DEV) * 2, N, H // 2, W // 2, 1)
                pieces = [engine.Piece(a), engine.Piece(b), engine.Piece(d, up=True)]
                cout = 64
            layer = engine.ConvLayer(mod)
            y, _, _ = engine.conv_forward(layer, pieces, ACT_LEAKY, 0.1, 0.0)
            kf = _lib.load().dn_last_kernel().decode()
            dy = torch.randn(N, H, W, cout, device=DEV)
            outs = [y]
            if case != "bn_pending_64_128":          # (the gradient w.r.t. a pre-BatchNorm tensor goes through the BN kernels)
                engine.conv_dgrad(layer, dy, N, H, W, pieces, (H, W))
                kd = _lib.load().dn_last_kernel().decode()
                outs += [pc.act.grad for pc in pieces]
            else:
                kd = kf
            dw = engine.conv_wgrad(layer, pieces, dy, (H, W))
            torch.cuda.synchronize()
            res[m] = (outs, kf, kd, dw)
    finally:
        engine.set_compute(prev)
    def _is(name):                                  # the three-piece mode has two kernels: the 4-wave one and the 8-wave form of it
        return ("wino_conv_kernel" in name and name.endswith(suffix)) or (mode == "f32x3" and "wino_conv8_kernel" in name)
    assert _is(res[mode][1]) and _is(res[mode][2]), res[mode][1:3]
    assert "wino_conv_kernel" in res["f32"][1] and res["f32"][1].endswith(", 0>")
    if mode == "bf16":
        assert torch.equal(res[mode][3], res["f32"][3])          # the weight gradient has no bf16-rounded variant
    else:
        rel = _rel_l2(res[mode][3], res["f32"][3])
        print("f32x3 %s dw: relative L2 %.3g" % (case, rel))
        assert rel <= 2e-6, (case, "dw", rel)
        close("%s dw" % case, res[mode][3], res["f32"][3], rtol=1e-4, atol_rel=1e-5)
    for i, (got, want) in enumerate(zip(res[mode][0], res["f32"][0])):
        rel = _rel_l2(got, want)
        mx = float((got - want).abs().max()) / (float(want.abs().max()) + 1e-30)
        print("%s %s[%d]: relative L2 %.3g, max error / magnitude %.3g" % (mode, case, i, rel, mx))
        if mode == "bf16":
            assert 1e-5 < rel <= 8e-3, (case, i, rel)        # (the lower bound: bf16 really was used)
            assert mx <= 3e-2, (case, i, mx)
        else:
            assert rel <= 2e-6, (case, i, rel)
            close("%s[%d]" % (case, i), got, want, rtol=1e-4, atol_rel=1e-5)


def test_bilinear_up2_matches_interpolate():
    x = rnd(2, 1, 5, 7, seed=1).requires_grad_()
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)[:, :, :9, :14]
    g = rnd(*ref.shape, seed=2)
    (ref * g).sum().backward()
    tape = engine.Tape(True)
    d = engine.Act(nhwc(x.detach()), 2, 5, 7, 1)
    up = engine.block_bilinear_up2(tape, d, (9, 14))
    close("bilinear", nchw(up.t), ref, rtol=1e-5, atol_rel=1e-6)
    up.grad = nhwc(g)
    tape.run_backward()
    close("bilinear_bwd", nchw(d.grad), x.grad, rtol=1e-5, atol_rel=1e-6)


def test_fused_adam_matches_torch_adam():
    from supervised_dispnet_amd.optim import FusedAdam
    torch.manual_seed(0)
    shapes = [(64, 3, 3, 3), (64,), (17,), (128, 64, 3, 3), (1, 16, 3, 3)]
    ref_p = [torch.randn(s).requires_grad_() for s in shapes]
    dev_p = [nn.Parameter(p.detach().clone().to(DEV)) for p in ref_p]
    opt_ref = torch.optim.Adam(ref_p, lr=1e-4, betas=(0.9, 0.999))
    opt = FusedAdam(dev_p, lr=1e-4, betas=(0.9, 0.999))
    for it in range(3):
        for p, q in zip(ref_p, dev_p):
            gr = torch.randn(p.shape, generator=torch.Generator().manual_seed(it * 100 + p.numel())) * 10 ** (it - 1)
            p.grad = gr.clone()
            q._dn_grad_view.copy_(gr.to(DEV))
        opt_ref.step()
        opt.step()
    for p, q in zip(ref_p, dev_p):
        close("adam", q.detach(), p.detach(), rtol=1e-6, atol_rel=1e-7)


@pytest.mark.parametrize("cmid,H,W", [(64, 16, 24), (128, 34, 26), (256, 16, 52)])
def test_bn_backward_sums_fused_into_the_input_gradient(cmid, H, W):
    """dn_conv_desc.bnb_*: the Winograd input-gradient kernels (4-wave and 8-wave) take the BatchNorm backward's column sums of the layer
    below in their epilogue.  conv-BN-ReLU -> conv-BN-ReLU backward with the fusion on and off: identical up to the summation order of the
    column sums (gradients of the first layer's BatchNorm pair, its convolution weight and its input)."""
    from supervised_dispnet_amd._lib import ACT_NONE  # noqa: F401
    torch.manual_seed(11)
    N, cin = 4, 32
    m1, b1 = nn.Conv2d(cin, cmid, 3, 1, 1).to(DEV), nn.BatchNorm2d(cmid).to(DEV)
    m2, b2 = nn.Conv2d(cmid, cmid, 3, 1, 1).to(DEV), nn.BatchNorm2d(cmid).to(DEV)
    with torch.no_grad():
        b1.weight.uniform_(0.5, 1.5); b1.bias.uniform_(-0.3, 0.3)
    x = torch.randn(N, H, W, cin, device=DEV)
    g = torch.randn(N, H, W, cmid, device=DEV)
    res = {}
    prev = engine.BN_SUMS_FUSION
    try:
        for fused in (True, False):
            engine.BN_SUMS_FUSION = fused
            for bn in (b1, b2):
                bn.reset_running_stats()
            tape, sink = engine.Tape(True), engine.GradSink()
            xa = engine.Act(x, N, H, W, cin)
            y1 = engine.block_conv_bn(tape, sink, engine.Piece(xa), engine.ConvLayer(m1), b1, True)
            y2 = engine.block_conv_bn(tape, sink, engine.Piece(y1), engine.ConvLayer(m2), b2, True)
            out = engine.block_bn_relu(tape, y2)
            out.grad = g.clone()
            tape.run_backward()
            engine.join_side_stream()
            torch.cuda.synchronize()
            kd = _lib.load().dn_last_kernel().decode()
            res[fused] = (sink.get(b1.weight).clone(), sink.get(b1.bias).clone(), sink.get(m1.weight).clone(), xa.grad.clone(), kd)
    finally:
        engine.BN_SUMS_FUSION = prev
    for a, b, what in zip(res[True][:4], res[False][:4], ("dgamma", "dbeta", "dweight", "dx")):
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 2e-5 * scale, (what, float((a - b).abs().max()), scale)
    # the fusion really ran: with it on, no separate sums pass is recorded for the first layer
    prof_on = []
    engine.PROFILE = prof_on
    try:
        engine.BN_SUMS_FUSION = True
        tape, sink = engine.Tape(True), engine.GradSink()
        xa = engine.Act(x, N, H, W, cin)
        y1 = engine.block_conv_bn(tape, sink, engine.Piece(xa), engine.ConvLayer(m1), b1, True)
        y2 = engine.block_conv_bn(tape, sink, engine.Piece(y1), engine.ConvLayer(m2), b2, True)
        out = engine.block_bn_relu(tape, y2)
        out.grad = g.clone()
        tape.run_backward()
        torch.cuda.synchronize()
    finally:
        engine.PROFILE = None
        engine.BN_SUMS_FUSION = prev
    assert sum(1 for r in prof_on if r[4] == "dn_bn_relu_bwd_sums") == 0, [r[4] for r in prof_on]


@pytest.mark.parametrize("case", ["c512_16x52", "c256_32x104_bn", "cat768_8x26", "cat_193_small"])
def test_winograd_input_channel_split_for_small_grids(case):
    """dn_conv_desc.splitk_ws: with few 32-tile blocks (a 4-image shard of the metric's batch) the three-piece Winograd forward / input
    gradient splits the K axis over 2-8 blocks per tile; the last arrival sums the partial tiles in index order.  Same results as the
    unsplit launch up to fp32 summation order (the split changes where the partial sums are cut), incl. the batch-statistic partials and
    a K axis that crosses operand boundaries."""
    torch.manual_seed(13)
    N = 4
    if case == "c512_16x52":                       # (names kept from the first draft; the shapes are the <= 128-block grids the split takes)
        H, W, cins, cout, bn = 8, 26, [512], 512, False
    elif case == "c256_32x104_bn":
        H, W, cins, cout, bn = 16, 24, [256], 256, True
    elif case == "cat768_8x26":
        H, W, cins, cout, bn = 8, 26, [256, 512], 256, False
    else:
        H, W, cins, cout, bn = 16, 24, [128, 256, 1], 64, False    # 8 + 16 + 1 chunks in three splits: [0,9) [9,18) [18,25)
    mod = nn.Conv2d(sum(cins), cout, 3, 1, 1).to(DEV)
    layer = engine.ConvLayer(mod)
    acts = []
    for i, c in enumerate(cins):
        if c == 1:
            a = engine.Act(torch.rand(N, H // 2, W // 2, 1, device=DEV) * 2, N, H // 2, W // 2, 1)
        else:
            a = engine.Act(torch.randn(N, H, W, c, device=DEV), N, H, W, c)
            if bn and i == 0:
                a.scale = torch.rand(c, device=DEV) + 0.5
                a.shift = torch.rand(c, device=DEV) - 0.5
        acts.append(a)
    pieces = [engine.Piece(a, up=(a.C == 1)) for a in acts]
    dy = torch.randn(N, H, W, cout, device=DEV)
    res = {}
    prev = engine.SPLITK
    try:
        for split in (True, False):
            engine.SPLITK = split
            for a in acts:
                a.grad = None
            y, partial, _ = engine.conv_forward(layer, pieces, bn_stats=True)
            kf = _lib.load().dn_last_kernel().decode()
            engine.conv_dgrad(layer, dy, N, H, W, pieces, (H, W))
            torch.cuda.synchronize()
            res[split] = (y.clone(), partial.clone(), [a.grad.clone() for a in acts], kf)
    finally:
        engine.SPLITK = prev
    assert "wino_conv_kernel" in res[True][3] and res[True][3] == res[False][3]
    ys, yn = res[True][0], res[False][0]
    assert not torch.equal(ys, yn)                                   # the split really ran (different summation cuts)
    scale = float(yn.abs().max())
    assert float((ys - yn).abs().max()) <= 2e-5 * scale
    ps, pn = res[True][1], res[False][1]
    assert float((ps - pn).abs().max()) <= 1e-4 * float(pn.abs().max())
    for gs, gn in zip(res[True][2], res[False][2]):
        assert float((gs - gn).abs().max()) <= 2e-5 * float(gn.abs().max())
    # run it twice more: the self-resetting counters leave the workspace reusable
    engine.SPLITK = True
    try:
        y2, _, _ = engine.conv_forward(layer, pieces, bn_stats=True)
        y3, _, _ = engine.conv_forward(layer, pieces, bn_stats=True)
        torch.cuda.synchronize()
        assert torch.equal(y2, ys) and torch.equal(y3, ys)           # deterministic: the partial tiles are summed in index order
    finally:
        engine.SPLITK = prev


@pytest.mark.parametrize("case", ["upconv5_b4", "upconv4_b4", "res_1x1s2", "convT_leaky_affine"])
def test_direct_three_piece_k_split_for_small_grids(case):
    """dn_conv_desc.splitk_ws in the three-piece direct kernel: the 4x13 / 8x26 transposed convolutions of the decoder (and the small
    strided convolutions of the ResNet encoders) are 8-104 tiles with 64-256 chunks of K each; blockIdx.y splits the chunks, the last
    arrival sums the partial accumulators in index order.  Same results as the unsplit launch up to fp32 summation order, forward
    (all four phases of the transposed convolution) and input gradient; deterministic; the workspace is reusable."""
    torch.manual_seed(17)
    if case == "upconv5_b4":
        mod, N, H, W, tr = nn.ConvTranspose2d(512, 256, 4, 2, 1), 4, 4, 13, True
    elif case == "upconv4_b4":
        mod, N, H, W, tr = nn.ConvTranspose2d(256, 128, 4, 2, 1), 4, 8, 26, True
    elif case == "res_1x1s2":
        mod, N, H, W, tr = nn.Conv2d(512, 1024, 1, 2, 0), 2, 30, 40, False
    else:
        mod, N, H, W, tr = nn.ConvTranspose2d(128, 64, 4, 2, 1), 2, 8, 12, True
    mod = mod.to(DEV)
    layer = engine.ConvLayer(mod, transposed=tr)
    cin = mod.in_channels
    a = engine.Act(torch.randn(N, H, W, cin, device=DEV), N, H, W, cin)
    if case == "convT_leaky_affine":
        a.scale = torch.rand(cin, device=DEV) + 0.5
        a.shift = torch.rand(cin, device=DEV) - 0.5
    pieces = [engine.Piece(a, False)]
    res = {}
    prev = engine.SPLITK
    try:
        for split in (True, False):
            engine.SPLITK = split
            a.grad = None
            y, _, _ = engine.conv_forward(layer, pieces, act=ACT_LEAKY, p0=0.1)
            kf = _lib.load().dn_last_kernel().decode()
            OH, OW = y.shape[1], y.shape[2]
            dy = torch.randn(N, OH, OW, mod.out_channels, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
            engine.conv_dgrad(layer, dy, N, OH, OW, pieces, (H, W))
            kd = _lib.load().dn_last_kernel().decode()
            torch.cuda.synchronize()
            res[split] = (y.clone(), a.grad.clone(), kf, kd)
    finally:
        engine.SPLITK = prev
    assert "igemm_conv_x3_kernel" in res[True][2] and res[True][2] == res[False][2], res[True][2]
    assert "igemm_conv_x3_kernel" in res[True][3], res[True][3]
    for i in (0, 1):
        s_, n_ = res[True][i], res[False][i]
        assert not torch.equal(s_, n_)                                  # the split really ran (different summation cuts)
        assert float((s_ - n_).abs().max()) <= 2e-5 * float(n_.abs().max())
    # against the framework's CPU fp32 convolution
    x_cpu = a.t.permute(0, 3, 1, 2).cpu()
    if a.scale is not None:
        x_cpu = F.relu(x_cpu * a.scale.cpu().view(1, -1, 1, 1) + a.shift.cpu().view(1, -1, 1, 1))
    y_ref = F.leaky_relu(copy.deepcopy(mod).cpu()(x_cpu), 0.1)
    close(case + ":y", nchw(res[True][0]), y_ref)
    engine.SPLITK = True
    try:
        y2, _, _ = engine.conv_forward(layer, pieces, act=ACT_LEAKY, p0=0.1)
        y3, _, _ = engine.conv_forward(layer, pieces, act=ACT_LEAKY, p0=0.1)
        torch.cuda.synchronize()
        assert torch.equal(y2, res[True][0]) and torch.equal(y3, res[True][0])
    finally:
        engine.SPLITK = prev


@pytest.mark.parametrize("case", ["256_256_ks2", "512_256_ks4_bn"])
def test_winograd8_tail_split_of_the_last_partial_round(case):
    """8-wave Winograd kernel (one block per CU): a grid of 256 k + tail blocks costs k + 1 rounds; the tiles of the partial round are
    split along the input channels instead (dn_winograd8.hip, dn_conv_desc.splitk_ws).  320 blocks = 256 + 64: the 64 tail tiles are
    split 2 / 4 ways.  Same results as the unsplit launch up to fp32 summation order in the tail tiles and bit-identical elsewhere;
    against the framework's CPU convolution; deterministic; input gradient too."""
    torch.manual_seed(23)
    N, H, W = 4, 32, 160                             # 4 x 16 x 80 = 5120 tiles = 80 blocks of 64 tiles x 4 blocks of 64 channels
    cin, cout, bn = (256, 256, False) if case == "256_256_ks2" else (512, 256, True)
    mod = nn.Conv2d(cin, cout, 3, 1, 1).to(DEV)
    layer = engine.ConvLayer(mod)
    a = engine.Act(torch.randn(N, H, W, cin, device=DEV), N, H, W, cin)
    if bn:
        a.scale = torch.rand(cin, device=DEV) + 0.5
        a.shift = torch.rand(cin, device=DEV) - 0.5
    pieces = [engine.Piece(a, False)]
    dy = torch.randn(N, H, W, cout, device=DEV)
    res = {}
    prev = engine.SPLITK
    try:
        for split in (True, False):
            engine.SPLITK = split
            a.grad = None
            y, partial, _ = engine.conv_forward(layer, pieces, bn_stats=True)
            kf = _lib.load().dn_last_kernel().decode()
            if cin == cout:
                engine.conv_dgrad(layer, dy, N, H, W, pieces, (H, W))
            torch.cuda.synchronize()
            res[split] = (y.clone(), partial.clone(), a.grad.clone() if a.grad is not None else None, kf)
    finally:
        engine.SPLITK = prev
    assert "wino_conv8_kernel" in res[True][3] and res[True][3] == res[False][3], res[True][3]
    ys, yn = res[True][0], res[False][0]
    differs = (ys != yn).reshape(N * H * W, cout // 64, 64).any(dim=2)      # per (pixel, 64-channel slice): the unit a block owns
    frac = float(differs.float().mean())
    assert 0.05 < frac <= 0.21, frac                 # only the tail (64 of 320 blocks = 20 % of the output) sees other summation cuts
    assert float((ys - yn).abs().max()) <= 2e-5 * float(yn.abs().max())
    assert float((res[True][1] - res[False][1]).abs().max()) <= 1e-4 * float(res[False][1].abs().max())
    if res[True][2] is not None:
        assert float((res[True][2] - res[False][2]).abs().max()) <= 2e-5 * float(res[False][2].abs().max())
    x_cpu = a.t.permute(0, 3, 1, 2).cpu()
    if bn:
        x_cpu = F.relu(x_cpu * a.scale.cpu().view(1, -1, 1, 1) + a.shift.cpu().view(1, -1, 1, 1))
    close(case + ":y", nchw(ys), copy.deepcopy(mod).cpu()(x_cpu))
    engine.SPLITK = True
    try:
        y2, _, _ = engine.conv_forward(layer, pieces, bn_stats=True)
        y3, _, _ = engine.conv_forward(layer, pieces, bn_stats=True)
        torch.cuda.synchronize()
        assert torch.equal(y2, ys) and torch.equal(y3, ys)
    finally:
        engine.SPLITK = prev
