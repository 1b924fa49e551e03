"""BASELINE-size checks (32 x 128 x 416, the metric configuration) through size-independent properties -- the CPU oracle
would need minutes per case at this size, so the full-size guarantees are stated as identities the domain offers:

  * adjointness: <dy, conv(x; w)> = <dgrad(dy; w), x> = <wgrad(x, dy), w>  (the three kernels of a layer are mutually
    transposed linear maps; evaluated in fp64 on the host from the fp32 device results);
  * linearity of the forward kernel in x;
  * a subsampled direct check: a few hundred output pixels recomputed on the CPU from their receptive fields;
  * eval-mode batch independence and run-to-run determinism of the whole network;
  * one full training step: loss finite, every hot parameter receives a finite non-zero gradient and moves.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

from supervised_dispnet_amd import engine  # noqa: E402
from supervised_dispnet_amd._lib import ACT_NONE  # noqa: E402

DEV = torch.device("cuda:0")
N, H, W = 32, 128, 416

FULL_LAYERS = [
    # name            cin  cout  h    w    what it exercises
    ("conv1_2",        64,  64, 128, 416),   # Winograd fwd / dgrad / wgrad, shortest K (4 chunks), largest tensors (436 MB)
    ("conv2_2",       128, 128,  64, 208),   # Winograd, 2 x 2 (co, ci) blocks x 64 tile splits in the weight gradient
    ("conv3_3",       256, 256,  32, 104),   # Winograd, 6.5 rounds of blocks
    ("conv4_3",       512, 512,  16,  52),   # Winograd, weights larger than one XCD's L2
    ("conv5_3",       512, 512,   8,  26),   # Winograd, long K, fewer blocks than slots
    ("first_layer",     3,  64, 128, 416),   # stem forward kernel + thin weight gradient (3-channel operand)
    ("iconv0_like",    16,  16, 128, 416),   # 32-wide implicit-GEMM tiles forward / dgrad, thin weight gradient
    ("disp_head",      16,   1, 128, 416),   # direct head kernels
]


def dot64(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize("name,cin,cout,h,w", FULL_LAYERS, ids=[l[0] for l in FULL_LAYERS])
def test_full_size_layer_identities(name, cin, cout, h, w):
    g = torch.Generator(device="cpu").manual_seed(7)
    mod = nn.Conv2d(cin, cout, 3, 1, 1).to(DEV)
    with torch.no_grad():
        mod.weight.copy_((torch.rand(mod.weight.shape, generator=g) - 0.5) * (2.0 / (9 * cin) ** 0.5))
        mod.bias.zero_()
    layer = engine.ConvLayer(mod)
    gen = torch.Generator(device=DEV).manual_seed(11)
    x = torch.rand(N, h, w, cin, device=DEV, generator=gen) - 0.5
    dy = torch.rand(N, h, w, cout, device=DEV, generator=gen) - 0.5
    xa = engine.Act(x, N, h, w, cin)
    xa.needs_grad = True
    pieces = [engine.Piece(xa)]
    y, _, _ = engine.conv_forward(layer, pieces, ACT_NONE)
    engine.conv_dgrad(layer, dy, N, h, w, pieces, (h, w))
    dw = engine.conv_wgrad(layer, pieces, dy, (h, w))
    torch.cuda.synchronize()
    a = dot64(dy, y)
    b = dot64(xa.grad, x)
    c = dot64(dw, mod.weight.detach())
    scale = float(dy.double().norm() * y.double().norm())
    assert abs(a - b) <= 2e-6 * scale, "%s: <dy,conv(x)> %.9g vs <dgrad(dy),x> %.9g" % (name, a, b)
    assert abs(a - c) <= 2e-6 * scale, "%s: <dy,conv(x)> %.9g vs <wgrad,w> %.9g" % (name, a, c)
    # linearity in x
    x2 = torch.rand(N, h, w, cin, device=DEV, generator=gen) - 0.5
    y2, _, _ = engine.conv_forward(layer, [engine.Piece(engine.Act(x2, N, h, w, cin))], ACT_NONE)
    y3, _, _ = engine.conv_forward(layer, [engine.Piece(engine.Act(0.5 * x - 2.0 * x2, N, h, w, cin))], ACT_NONE)
    lin = (0.5 * y - 2.0 * y2 - y3).abs().max()
    assert float(lin) <= 2e-5 * float(y.abs().max() + 2 * y2.abs().max()), "%s: forward not linear (%.3g)" % (name, float(lin))
    # subsampled direct check against F.conv2d on the receptive fields (CPU fp32)
    r = np.random.RandomState(3)
    wc = mod.weight.detach().cpu()
    for _ in range(64):
        n, oy, ox = r.randint(N), r.randint(h), r.randint(w)
        y0, y1, x0, x1 = max(oy - 1, 0), min(oy + 2, h), max(ox - 1, 0), min(ox + 2, w)
        patch = torch.zeros(1, cin, 3, 3)
        patch[0, :, y0 - oy + 1:y1 - oy + 1, x0 - ox + 1:x1 - ox + 1] = x[n, y0:y1, x0:x1, :].permute(2, 0, 1).cpu()
        want = F.conv2d(patch, wc)[0, :, 0, 0]
        got = y[n, oy, ox, :].cpu()
        assert torch.allclose(got, want, rtol=2e-4, atol=2e-5 * float(want.abs().max() + 1e-6)), (name, n, oy, ox)


def test_full_size_network_properties():
    import supervised_dispnet_amd.loss_functions as LF
    import supervised_dispnet_amd.models as models
    from supervised_dispnet_amd.functional import reciprocal
    from supervised_dispnet_amd.optim import FusedAdam
    torch.manual_seed(0)
    net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
    net.init_weights(use_pretrained_weights=False)
    net.to(DEV)
    g = torch.Generator().manual_seed(0)
    img = ((torch.rand(N, 3, H, W, generator=g) - 0.5) / 0.5).to(DEV)
    depth = torch.rand(N, H, W, generator=g) * 79.0 + 1.0
    gt = (depth * (torch.rand(N, H, W, generator=g) < 0.05).float()).to(DEV)
    # eval mode: every sample is independent of the rest of the batch, and the run is deterministic
    net.eval()
    with torch.no_grad():
        full = net(img)
        again = net(img)
        part = net(img[5:9].contiguous())
    assert full.shape == (N, 1, H, W) and torch.equal(full, again)
    assert torch.allclose(full[5:9], part, rtol=1e-5, atol=1e-6)
    assert float(full.min()) > 0.0099 and float(full.max()) < 10.0101          # alpha * sigmoid + beta range
    # one training step at the metric configuration
    net.train()
    opt = FusedAdam(net._hot_parameters(), lr=1e-4, betas=(0.9, 0.999), production_order=net._grad_production_order())
    before = opt.arena.flat_p.clone()
    disps = net(img)
    assert [tuple(d.shape) for d in disps] == [(N, 1, 128, 416), (N, 1, 64, 208), (N, 1, 32, 104), (N, 1, 16, 52)]
    loss = LF.l1_loss(gt, [reciprocal(d) for d in disps], "kitti")
    opt.zero_grad()
    loss.backward()
    assert np.isfinite(loss.item())
    gflat = opt.arena.flat_g
    assert torch.isfinite(gflat).all()
    for p_ in opt.arena.params:
        # (the bias of a conv that feeds BatchNorm has an exactly-zero gradient: the batch mean removes it)
        assert p_.dim() == 1 or float(p_._dn_grad_view.abs().max()) > 0.0
    assert sum(float(p_._dn_grad_view.abs().max()) > 0.0 for p_ in opt.arena.params) >= 0.75 * len(opt.arena.params)
    opt.step()
    moved = (opt.arena.flat_p - before).abs()
    assert torch.isfinite(opt.arena.flat_p).all() and float(moved.max()) <= 1.001e-4 and float(moved.max()) > 5e-5   # |Adam step 1| = lr (up to fp32 rounding of p - step)


def test_two_stream_backward_is_bitwise_the_single_stream_one(monkeypatch):
    """engine.WGRAD_STREAM: the weight gradients run on a side HIP stream next to the input gradients.  Same kernels, same
    arguments, only the schedule differs -> the whole gradient arena must be bit-identical to the single-stream run (a missing
    fence or a buffer recycled under a running kernel would show up here), at the metric's size, over a few steps."""
    import supervised_dispnet_amd.loss_functions as LF
    import supervised_dispnet_amd.models as models
    from supervised_dispnet_amd.functional import reciprocal
    from supervised_dispnet_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(1)
    img = ((torch.rand(N, 3, H, W, generator=g) - 0.5) / 0.5).to(DEV)
    gt = ((torch.rand(N, H, W, generator=g) * 79.0 + 1.0) * (torch.rand(N, H, W, generator=g) < 0.05).float()).to(DEV)
    grads, params = {}, {}
    for mode in ("0", "1"):
        monkeypatch.setattr(engine, "_WGRAD_STREAM_MODE", mode)
        monkeypatch.setattr(engine, "WGRAD_STREAM", mode != "0")
        torch.manual_seed(0)
        net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
        net.init_weights(use_pretrained_weights=False)
        net.to(DEV).train()
        opt = FusedAdam(net._hot_parameters(), lr=1e-4, betas=(0.9, 0.999), production_order=net._grad_production_order())
        for _ in range(3):
            loss = LF.l1_loss(gt, [reciprocal(d) for d in net(img)], "kitti")
            opt.zero_grad()
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        grads[mode], params[mode] = opt.arena.flat_g.clone(), opt.arena.flat_p.clone()
    assert torch.equal(grads["0"], grads["1"]) and torch.equal(params["0"], params["1"])
