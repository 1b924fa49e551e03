"""Fused DORN head (dn_ord_head_fwd / dn_ord_head_bwd: Dropout2d mask -> 1x1 conv 16 -> 2K -> clamp -> pair softmax, logits never
in HBM) against the reference's module sequence restated on the CPU (models/Disp_vgg_BN_DORN.py:112-114,191-227) and against the
unfused HIP path (conv kernel + dn_ordinal_fwd/bwd)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

from oracle import nets as ON  # noqa: E402
from supervised_dispnet_amd import engine  # noqa: E402

DEV = torch.device("cuda:0")


@pytest.mark.parametrize("N,H,W,K,use_mask", [(2, 16, 24, 8, True), (3, 8, 16, 71, False), (2, 32, 48, 80, True), (1, 8, 8, 17, True)])
def test_fused_ord_head_matches_cpu_reference(N, H, W, K, use_mask):
    g = torch.Generator().manual_seed(K * 100 + N)
    conv = nn.Conv2d(16, 2 * K, 1)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.4)
        conv.bias.copy_(torch.randn(2 * K, generator=g) * 0.3)
    x = torch.randn(N, 16, H, W, generator=g)
    mask = ((torch.rand(N, 16, generator=g) < 0.5).float() * 2.0) if use_mask else None
    G = torch.randn(N, K, H, W, generator=g)
    # ---- CPU: the reference's module sequence
    xc = x.clone().requires_grad_(True)
    wc, bc = conv.weight.detach().clone().requires_grad_(True), conv.bias.detach().clone().requires_grad_(True)
    xin = xc * mask.view(N, 16, 1, 1) if use_mask else xc
    dec_c, ord_c = ON.ordinal_regression(F.conv2d(xin, wc, bc))
    (ord_c * G).sum().backward()
    # ---- HIP, fused
    conv_d = conv.to(DEV)
    xa = engine.Act(x.permute(0, 2, 3, 1).contiguous().to(DEV), N, H, W, 16)
    assert engine.ord_head_fusable(xa, K)
    tape, sink = engine.Tape(True), engine.GradSink()
    o, d = engine.block_ord_head(tape, sink, xa, conv_d, mask.to(DEV) if use_mask else None)
    engine.seed_grad(o, G.to(DEV))
    tape.run_backward()
    torch.cuda.synchronize()
    assert tuple(o.t.shape) == (N, K, H, W) and d.t.dtype == torch.int64 and tuple(d.t.shape) == (N, 1, H, W)
    np.testing.assert_allclose(o.t.cpu().numpy(), ord_c.detach().numpy(), rtol=2e-5, atol=2e-6)
    flips = (d.t.cpu() != dec_c)
    near = ((ord_c.detach() - 0.5).abs() < 1e-5).sum(1, keepdim=True) > 0
    assert not (flips & ~near).any()                                    # decode differs only where some P is within rounding of 0.5
    sc = lambda t: float(t.abs().max()) + 1e-30
    dx = xa.grad.permute(0, 3, 1, 2).cpu()
    assert float((dx - xc.grad).abs().max()) <= 2e-5 * sc(xc.grad)
    dw, db = sink.get(conv_d.weight).cpu(), sink.get(conv_d.bias).cpu()
    assert float((dw - wc.grad).abs().max()) <= 5e-5 * sc(wc.grad), (float((dw - wc.grad).abs().max()), sc(wc.grad))
    assert float((db - bc.grad).abs().max()) <= 5e-5 * sc(bc.grad)
    # accumulate form: a second backward into the existing x.grad doubles it
    tape2, sink2 = engine.Tape(True), engine.GradSink()
    o2, _ = engine.block_ord_head(tape2, sink2, xa, conv_d, mask.to(DEV) if use_mask else None)
    engine.seed_grad(o2, G.to(DEV))
    tape2.run_backward()
    assert torch.allclose(xa.grad.permute(0, 3, 1, 2).cpu(), 2 * dx, rtol=1e-6, atol=1e-7 * sc(dx))


def test_fused_head_equals_unfused_network_path():
    """The whole Disp_vgg_BN_DORN (K = 80, 2 x 64 x 96): fused head vs the r01 path (conv_ord logits in HBM + dn_ordinal_*)."""
    import supervised_dispnet_amd.loss_functions as LF
    import supervised_dispnet_amd.models as models
    import supervised_dispnet_amd.utils as U
    from oracle import detgen
    res = {}
    for fused in (True, False):
        net = models.Disp_vgg_BN_DORN(datasets="kitti", ordinal_c=80, with_classifier=False)
        detgen.fill_state_dict(net.state_dict(), "ordhead")
        net.fused_head = fused
        net.to(DEV).train()
        net._dropout_mask = (detgen.bernoulli((2, 16), "ordhead:drop", 0.5).float() * 2.0).to(DEV)
        x = detgen.image_batch(2, 64, 96, "ordhead:x").to(DEV)
        gt = detgen.sparse_depth(2, 64, 96, "ordhead:gt", density=0.3).to(DEV)
        tgt = U.get_labels_sid(gt, ordinal_c=80, dataset="kitti")
        dec, ordc = net(x)
        loss = LF.DORN_loss(gt, ordc, tgt, "kitti")
        loss.backward()
        res[fused] = (dec, ordc, loss.item(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    assert tuple(res[True][1].shape) == (2, 80, 64, 96) and res[True][1].is_contiguous()
    np.testing.assert_allclose(res[True][2], res[False][2], rtol=1e-5)
    assert torch.allclose(res[True][1], res[False][1], rtol=1e-4, atol=1e-6)
    assert (res[True][0] != res[False][0]).float().mean() < 1e-3
    for k, gf in res[True][3].items():
        gu = res[False][3][k]
        err = float((gf - gu).norm() / (gu.norm() + 1e-30))
        assert err < 2e-3, (k, err)
