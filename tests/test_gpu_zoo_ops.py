"""The pointwise pieces added for FCRN / the ASPP nets, each against its PyTorch-CPU fp32 definition (pytest -m gpu):
dilated convolutions (forward, input gradient, weight gradient), an up-projection branch (four un-padded 3x3 / 2x3 / 3x2 / 2x2 convolutions interleaved by parity)
as one composite transposed convolution, ceil-mode 3x3 / stride-2 max-pool, bilinear resize with
align_corners (forward bit-close to ATen, backward against autograd), batch statistics of a materialised tensor."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

from supervised_dispnet_amd import _lib, engine  # noqa: E402

DEV = torch.device("cuda:0")


def _nchw(t):
    return t.permute(0, 3, 1, 2).cpu()


@pytest.mark.parametrize("cin,cout,dil,H,W", [(64, 64, 2, 16, 24), (32, 48, 4, 9, 13), (256, 1, 6, 9, 13), (64, 1, 24, 17, 27)])
def test_dilated_conv_fwd_dgrad_wgrad(cin, cout, dil, H, W):
    torch.manual_seed(1)
    N = 2
    mod = nn.Conv2d(cin, cout, 3, 1, padding=dil, dilation=dil).to(DEV)
    layer = engine.ConvLayer(mod)
    x = torch.randn(N, H, W, cin, device=DEV)
    xa = engine.Act(x, N, H, W, cin)
    y, _, _ = engine.conv_forward(layer, [engine.Piece(xa)])
    assert "wino" not in _lib.load().dn_last_kernel().decode()
    xr = _nchw(x).requires_grad_(True)
    mc = nn.Conv2d(cin, cout, 3, 1, padding=dil, dilation=dil)
    mc.load_state_dict({k: v.cpu() for k, v in mod.state_dict().items()})
    yr = mc(xr)
    np.testing.assert_allclose(_nchw(y).numpy(), yr.detach().numpy(), rtol=1e-4, atol=1e-4)
    dy = torch.randn(N, H, W, cout, device=DEV)
    yr.backward(_nchw(dy))
    engine.conv_dgrad(layer, dy, N, H, W, [engine.Piece(xa)], (H, W))
    dw = engine.conv_wgrad(layer, [engine.Piece(xa)], dy, (H, W))
    torch.cuda.synchronize()
    np.testing.assert_allclose(_nchw(xa.grad).numpy(), xr.grad.numpy(), rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(dw.cpu().numpy(), mc.weight.grad.numpy(), rtol=1e-4, atol=2e-4 * float(mc.weight.grad.abs().max()))


def test_upprojection_branch_is_a_composite_transposed_conv():
    """models/FCRN.py:74-107: four un-padded convolutions (3x3, 2x3, 3x2, 2x2) over hand-padded inputs, interleaved by parity == one
    ConvTranspose2d(6x6, stride 2, padding 2) on the composite weight + the four per-phase biases; forward, input gradient and the four
    weight / bias gradients against the reference's own formulation on the CPU."""
    torch.manual_seed(2)
    N, H, W, cin, cout = 2, 7, 9, 32, 16
    kss = ((3, 3), (2, 3), (3, 2), (2, 2))
    pads = ((1, 1, 1, 1), (1, 1, 1, 0), (1, 0, 1, 1), (1, 0, 1, 0))
    convs = [nn.Conv2d(cin, cout, ks).to(DEV) for ks in kss]
    comp = engine._CompositeConvT(convs)
    layer = engine.ConvLayer(comp, transposed=True)
    x = torch.randn(N, H, W, cin, device=DEV)
    xa = engine.Act(x, N, H, W, cin)
    comp.assemble()
    o_t, _, _ = engine.conv_forward(layer, [engine.Piece(xa)])
    assert tuple(o_t.shape) == (N, 2 * H, 2 * W, cout)
    bias4 = torch.stack([m.bias.detach() for m in convs]).contiguous()
    _lib.call("dn_phase_bias_add", o_t.data_ptr(), N, 2 * H, 2 * W, cout, bias4.data_ptr(), engine._stream())
    cpu = [nn.Conv2d(cin, cout, ks) for ks in kss]
    for c, m in zip(cpu, convs):
        c.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    xr = _nchw(x).requires_grad_(True)
    o = [c(F.pad(xr, pd)) for c, pd in zip(cpu, pads)]
    top = torch.stack((o[0], o[1]), dim=-1).reshape(N, cout, H, 2 * W)
    bot = torch.stack((o[2], o[3]), dim=-1).reshape(N, cout, H, 2 * W)
    ref = torch.stack((top, bot), dim=-2).reshape(N, cout, 2 * H, 2 * W)
    np.testing.assert_allclose(_nchw(o_t).numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-4)
    g = torch.randn(N, 2 * H, 2 * W, cout, device=DEV)
    ref.backward(_nchw(g))
    engine.conv_dgrad(layer, g, N, 2 * H, 2 * W, [engine.Piece(xa)], (H, W))
    dwt = engine.conv_wgrad(layer, [engine.Piece(xa)], g, (2 * H, 2 * W))
    engine.join_side_stream()
    sink = engine.GradSink()
    comp.scatter_grad(dwt, sink)
    ws = torch.empty(_lib.load().dn_phase_colsum_workspace_bytes(cout) // 4, device=DEV)
    db4 = torch.empty((4, cout), device=DEV)
    _lib.call("dn_phase_colsum", g.data_ptr(), N, H, W, cout, ws.data_ptr(), db4.data_ptr(), engine._stream())
    torch.cuda.synchronize()
    np.testing.assert_allclose(_nchw(xa.grad).numpy(), xr.grad.numpy(), rtol=1e-4, atol=2e-4)
    for k, (c, m) in enumerate(zip(cpu, convs)):
        np.testing.assert_allclose(sink.get(m.weight).cpu().numpy(), c.weight.grad.numpy(), rtol=1e-4, atol=2e-4 * float(c.weight.grad.abs().max()))
        np.testing.assert_allclose(db4[k].cpu().numpy(), c.bias.grad.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("H,W,ceil", [(32, 48, True), (33, 47, True), (32, 48, False), (7, 9, True)])
def test_maxpool3s2_ceil_mode(H, W, ceil):
    torch.manual_seed(3)
    N, Cn = 2, 8
    x = torch.randn(N, H, W, Cn, device=DEV)
    tape = engine.Tape(True)
    xa = engine.Act(x, N, H, W, Cn)
    out = engine.block_maxpool3s2(tape, xa, ceil_mode=ceil)
    xr = _nchw(x).requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1, ceil_mode=ceil)
    assert tuple(out.t.shape[1:3]) == tuple(yr.shape[2:])
    assert torch.equal(_nchw(out.t), yr.detach())
    g = torch.randn_like(out.t)
    out.grad = g
    yr.backward(_nchw(g))
    tape.run_backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(_nchw(xa.grad).numpy(), xr.grad.numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("ih,iw,oh,ow,align", [(32, 48, 64, 96, True), (17, 27, 64, 96, True), (9, 13, 64, 96, True), (32, 48, 64, 96, False),
                                              (64, 96, 33, 47, True), (5, 7, 5, 7, True), (1, 1, 4, 6, True)])
def test_resize_bilinear_matches_aten(ih, iw, oh, ow, align):
    torch.manual_seed(4)
    N = 3
    x = torch.randn(N, ih, iw, 1, device=DEV)
    tape = engine.Tape(True)
    xa = engine.Act(x, N, ih, iw, 1)
    out = engine.block_resize_bilinear(tape, xa, (oh, ow), align_corners=align)
    xr = x.view(N, 1, ih, iw).cpu().requires_grad_(True)
    yr = F.interpolate(xr, size=(oh, ow), mode="bilinear", align_corners=align)
    np.testing.assert_allclose(out.t.view(N, 1, oh, ow).cpu().numpy(), yr.detach().numpy(), rtol=1e-5, atol=1e-6)
    g = torch.randn(N, oh, ow, 1, device=DEV)
    out.grad = g
    yr.backward(g.view(N, 1, oh, ow).cpu())
    tape.run_backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(xa.grad.view(N, 1, ih, iw).cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("rows,Cn", [(1000, 64), (128 * 7, 16), (77, 8), (5000, 130)])
def test_bn_stats_of_a_materialised_tensor(rows, Cn):
    torch.manual_seed(5)
    x = torch.randn(rows, Cn, device=DEV) * 3 + 1.5
    lib = _lib.load()
    prow = lib.dn_bn_stats_rows(rows)
    partial = torch.empty((prow, Cn, 2), device=DEV)
    _lib.call("dn_bn_stats_partial", x.data_ptr(), rows, Cn, partial.data_ptr(), engine._stream())
    bn = nn.BatchNorm2d(Cn).to(DEV)
    y = engine.Act(x.view(1, rows, 1, Cn), 1, rows, 1, Cn)
    engine._bn_pending(y, bn, partial, prow, True)
    torch.cuda.synchronize()
    xd = x.double().cpu()
    np.testing.assert_allclose(y.mean.cpu().numpy(), xd.mean(0).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(y.invstd.cpu().numpy(), (1 / torch.sqrt(xd.var(0, unbiased=False) + bn.eps)).numpy(), rtol=1e-5)


def test_bn_finalize_keeps_the_variance_of_a_channel_with_a_large_offset():
    """|mean| >> std.  The merge of the per-tile (sum, M2) pairs is expanded about the first tile's mean, not about 0, so
    mean^2 / var = 1e10 (fp32 data: offset 4096, spread 0.04) loses nothing: the finalize step is exact to fp64 rounding on ITS inputs,
    and what the test compares against is the fp64 statistics of the same fp32 tensor."""
    torch.manual_seed(11)
    rows, Cn = 128 * 301 + 17, 8
    off = torch.tensor([4096.0, -4096.0, 1024.0, 0.0, 3.0e4, 1.0, -512.0, 65536.0], device=DEV)
    spread = torch.tensor([0.04, 0.04, 0.02, 1.0, 0.5, 1e-3, 0.01, 1.0], device=DEV)
    drift = torch.linspace(-1, 1, rows, device=DEV).view(rows, 1)                      # tile means differ: the between-tile term is live
    x = (off + spread * (torch.randn(rows, Cn, device=DEV) + 2.0 * drift)).contiguous()
    lib = _lib.load()
    prow = lib.dn_bn_stats_rows(rows)
    partial = torch.empty((prow, Cn, 2), device=DEV)
    _lib.call("dn_bn_stats_partial", x.data_ptr(), rows, Cn, partial.data_ptr(), engine._stream())
    bn = nn.BatchNorm2d(Cn, eps=1e-12).to(DEV)
    y = engine.Act(x.view(1, rows, 1, Cn), 1, rows, 1, Cn)
    engine._bn_pending(y, bn, partial, prow, True)
    torch.cuda.synchronize()
    xd = x.double().cpu()
    var = xd.var(0, unbiased=False)
    np.testing.assert_allclose(y.mean.cpu().numpy(), xd.mean(0).numpy(), rtol=5e-7, atol=1e-7)     # (atol: the zero-offset channel, |mean| << std)
    np.testing.assert_allclose(y.invstd.cpu().numpy(), (1 / torch.sqrt(var + bn.eps)).numpy(), rtol=2e-3)   # (the tiles' own fp32 sums bound this)
