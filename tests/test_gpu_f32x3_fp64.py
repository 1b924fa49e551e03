"""The product's DEFAULT arithmetic -- fp32 products formed on the bf16 matrix cores from three exact bf16 pieces per operand
(DN_COMPUTE_F32X3) -- pinned to fp64 at the METRIC's sizes, not at a microbenchmark's.  pytest -m gpu.

  * forward, input gradient AND weight gradient (dn::wino_conv_kernel<..., 3>, dn::wino_wgrad_x3_kernel) against an fp64 evaluation
    of the same fp32 inputs at 64 -> 64 @ 32 x 128 x 416 (the weight gradient sums 425 984 tiles) and 512 -> 512 @ 32 x 16 x 52
    (K = 4608): error <= 1.5 x the fp32-instruction kernel's on the same data (+ 1e-7 of the magnitude).  The fp64 yardstick is
    evaluated on SAMPLED outputs (a few thousand output pixels x all channels; 8 input channels x all output channels x 9 taps for
    the weight gradient, each a full 1.7 M-pixel reduction) with plain torch fp64 gathers + matmul: test infrastructure.
  * an exponent-range sweep (operands x 1e-30 ... 1e+30): the same relative bound at every scale -- bf16 pieces have fp32's
    exponent range, nothing is rescaled anywhere.  Below ~2^-110 the third piece of an operand becomes a bf16 denormal: outside
    the validated range.
  * non-finite inputs: a +/-Inf or NaN input pixel makes every output of the 2 x 2 output tiles whose 4 x 4 input patch contains it
    non-finite (NaN through Inf - Inf in the Winograd input transform in EITHER mode, and through x - bf16(x) in the split) and
    leaves every other output bit-identical to the clean run; both modes agree on the set.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

from supervised_dispnet_amd import _lib, engine  # noqa: E402
from supervised_dispnet_amd._lib import ACT_NONE  # noqa: E402

DEV = torch.device("cuda:0")


def _run(mode, mod, x, dy, N, H, W, cin, cout, what):
    """One launch of the Winograd kernel under test in `mode`; returns (tensor, kernel name)."""
    engine.set_compute(mode)
    layer = engine.ConvLayer(mod)
    xa = engine.Act(x, N, H, W, cin)
    if what == "fwd":
        out, _, _ = engine.conv_forward(layer, [engine.Piece(xa)], ACT_NONE)
    elif what == "dgrad":
        engine.conv_dgrad(layer, dy, N, H, W, [engine.Piece(xa)], (H, W))
        out = xa.grad
    else:
        out = engine.conv_wgrad(layer, [engine.Piece(xa)], dy, (H, W))
    name = _lib.load().dn_last_kernel().decode()
    torch.cuda.synchronize()
    return out, name


def _pad_nhwc64(t):
    return torch.nn.functional.pad(t.double(), (0, 0, 1, 1, 1, 1))        # zero halo on H and W of [N,H,W,C]


def _ref_fwd_samples(x, w, b, idx):
    """fp64 y[n,h,w,:] at sampled pixels idx = (n, h, w) index tensors: sum_{r,s,ci} x[n,h+r-1,w+s-1,ci] w[co,ci,r,s] + b."""
    xp = _pad_nhwc64(x)
    n, h, ww = idx
    acc = torch.zeros((n.numel(), w.shape[0]), dtype=torch.float64, device=x.device)
    w64 = w.double()
    for r in range(3):
        for s in range(3):
            acc += xp[n, h + r, ww + s] @ w64[:, :, r, s].t()
    return acc + (b.double() if b is not None else 0)


def _ref_dgrad_samples(dy, w, idx):
    """fp64 dx[n,h,w,:] = sum_{r,s,co} dy[n,h-r+1,w-s+1,co] w[co,ci,r,s]."""
    dp = _pad_nhwc64(dy)
    n, h, ww = idx
    acc = torch.zeros((n.numel(), w.shape[1]), dtype=torch.float64, device=dy.device)
    w64 = w.double()
    for r in range(3):
        for s in range(3):
            acc += dp[n, h + 2 - r, ww + 2 - s] @ w64[:, :, r, s]
    return acc


def _ref_wgrad_samples(x, dy, ci_idx):
    """fp64 dW[:, ci_idx, r, s] = sum over ALL pixels of dy[n,h,w,co] x[n,h+r-1,w+s-1,ci]."""
    N, H, W, _ = x.shape
    xp = _pad_nhwc64(x[..., ci_idx])
    d = dy.double().reshape(-1, dy.shape[-1])
    out = torch.zeros((dy.shape[-1], len(ci_idx), 3, 3), dtype=torch.float64, device=x.device)
    for r in range(3):
        for s in range(3):
            out[:, :, r, s] = d.t() @ xp[:, r:r + H, s:s + W].reshape(-1, len(ci_idx))
    return out


def _errs(got, ref):
    e = (got.double() - ref).abs()
    return float(e.max()), float(e.norm() / (ref.norm() + 1e-300)), float(ref.abs().max())


SHAPES = [("64_64_at_32x128x416", 32, 128, 416, 64, 64), ("512_512_at_32x16x52", 32, 16, 52, 512, 512)]


@pytest.mark.parametrize("what", ["fwd", "dgrad", "wgrad"])
@pytest.mark.parametrize("shape", SHAPES, ids=[s[0] for s in SHAPES])
def test_three_piece_kernels_vs_fp64_at_metric_size(shape, what):
    _tag, N, H, W, cin, cout = shape
    torch.manual_seed(21)
    mod = nn.Conv2d(cin, cout, 3, 1, 1).to(DEV)
    x = torch.rand(N, H, W, cin, device=DEV) * 2.0                     # non-negative like a post-ReLU activation: no cancellation helps
    dy = torch.randn(N, H, W, cout, device=DEV) * 0.05
    g = torch.Generator().manual_seed(5)
    P = 4096
    idx = (torch.randint(0, N, (P,), generator=g).to(DEV), torch.randint(0, H, (P,), generator=g).to(DEV),
           torch.randint(0, W, (P,), generator=g).to(DEV))
    idx[1][:64] = 0; idx[1][64:128] = H - 1; idx[2][128:192] = 0; idx[2][192:256] = W - 1     # borders are sampled for sure
    w, b = mod.weight.detach(), mod.bias.detach()
    ci_idx = list(range(0, cin, max(cin // 8, 1)))[:8]
    if what == "fwd":
        ref = _ref_fwd_samples(x, w, b, idx)
    elif what == "dgrad":
        ref = _ref_dgrad_samples(dy, w, idx)
    else:
        ref = _ref_wgrad_samples(x, dy, ci_idx)
    prev = engine.compute_mode()
    res = {}
    try:
        for mode in ("f32", "f32x3"):
            out, name = _run(mode, mod, x, dy, N, H, W, cin, cout, what)
            if what == "wgrad":
                assert ("wino_wgrad_x3" in name) == (mode == "f32x3") and "wino_wgrad" in name, name      # (x3: the 64 x 64 x 16 or the 128 x 64 x 8-position block)
                got = out[:, ci_idx]
            else:
                assert ("wino_conv8_kernel" in name or ("wino_conv_kernel" in name and name.endswith(", 3>"))) if mode == "f32x3" else ("wino_conv_kernel" in name and name.endswith(", 0>")), name
                got = out[idx[0], idx[1], idx[2]]
            res[mode] = _errs(got, ref)
    finally:
        engine.set_compute(prev)
    if what == "wgrad":
        # the reference's own arithmetic on the same data: PyTorch-CPU fp32 weight gradient (what `loss.backward()` computes there)
        xc = x.permute(0, 3, 1, 2).cpu().contiguous()
        dyc = dy.permute(0, 3, 1, 2).cpu().contiguous()
        wc = torch.nn.functional.conv2d  # noqa: F841
        dw_cpu = torch.nn.grad.conv2d_weight(xc, (cout, cin, 3, 3), dyc, stride=1, padding=1)
        res["cpu"] = _errs(dw_cpu[:, ci_idx].to(DEV), ref)
        print("%s wgrad vs fp64: PyTorch-CPU fp32 max %.3g relL2 %.3g" % (shape[0], res["cpu"][0], res["cpu"][1]))
    (m3, l3, mag), (m0, l0, _) = res["f32x3"], res["f32"]
    print("%s %s vs fp64: three-piece max %.3g relL2 %.3g | fp32 instruction max %.3g relL2 %.3g | magnitude %.3g" % (shape[0], what, m3, l3, m0, l0, mag))
    assert l0 < 5e-6 and l3 < 5e-6, "an fp32 result must sit at fp32 round-off of the fp64 value"
    assert l3 <= 1.5 * l0 + 1e-8, (l3, l0)
    assert m3 <= 1.5 * m0 + 1e-7 * mag, (m3, m0, mag)


@pytest.mark.parametrize("scale", [1e-30, 1e-20, 1e-10, 1.0, 1e10, 1e20, 1e30])
def test_three_piece_exponent_range(scale):
    """x (forward, weight gradient) resp. dy (input gradient) scaled by `scale`: the relative error against fp64 keeps the bound of
    the unscaled case in both modes (the result scales along, 1e-30 * O(100) ... 1e+30 * O(100) stay inside fp32's range)."""
    N, H, W, cin, cout = 4, 32, 48, 128, 64
    torch.manual_seed(31)
    mod = nn.Conv2d(cin, cout, 3, 1, 1, bias=False).to(DEV)
    x1 = torch.randn(N, H, W, cin, device=DEV)
    dy1 = torch.randn(N, H, W, cout, device=DEV)
    g = torch.Generator().manual_seed(6)
    P = 2048
    idx = (torch.randint(0, N, (P,), generator=g).to(DEV), torch.randint(0, H, (P,), generator=g).to(DEV),
           torch.randint(0, W, (P,), generator=g).to(DEV))
    w = mod.weight.detach()
    ci_idx = list(range(0, cin, 16))
    prev = engine.compute_mode()
    try:
        for what in ("fwd", "dgrad", "wgrad"):
            x = x1 * scale if what != "dgrad" else x1
            dy = dy1 * scale if what == "dgrad" else dy1
            if what == "fwd":
                ref = _ref_fwd_samples(x, w, None, idx)
            elif what == "dgrad":
                ref = _ref_dgrad_samples(dy, w, idx)
            else:
                ref = _ref_wgrad_samples(x, dy, ci_idx)
            l = {}
            for mode in ("f32", "f32x3"):
                out, name = _run(mode, mod, x, dy, N, H, W, cin, cout, what)
                got = out[:, ci_idx] if what == "wgrad" else out[idx[0], idx[1], idx[2]]
                assert torch.isfinite(got).all(), (what, mode, scale)
                l[mode] = _errs(got, ref)[1]
            print("scale %.0e %s: relative L2 vs fp64 three-piece %.3g, fp32 instruction %.3g" % (scale, what, l["f32x3"], l["f32"]))
            assert l["f32x3"] <= 1.5 * l["f32"] + 1e-8 and l["f32x3"] < 5e-6, (what, scale, l)
    finally:
        engine.set_compute(prev)


@pytest.mark.parametrize("bad", [float("inf"), float("-inf"), float("nan")])
def test_non_finite_inputs_are_contained_to_their_tiles(bad):
    N, H, W, cin, cout = 2, 16, 24, 64, 64
    torch.manual_seed(41)
    mod = nn.Conv2d(cin, cout, 3, 1, 1).to(DEV)
    x = torch.randn(N, H, W, cin, device=DEV)
    py, px, pc = 7, 10, 5                                              # an interior pixel
    xb = x.clone()
    xb[1, py, px, pc] = bad
    prev = engine.compute_mode()
    try:
        sets = {}
        for mode in ("f32", "f32x3"):
            clean, _ = _run(mode, mod, x, None, N, H, W, cin, cout, "fwd")
            dirty, name = _run(mode, mod, xb, None, N, H, W, cin, cout, "fwd")
            assert "wino_conv" in name
            nf = ~torch.isfinite(dirty)
            # every output whose 3x3 receptive field holds the pixel is non-finite, on every output channel
            assert bool(nf[1, py - 1:py + 2, px - 1:px + 2].all()), mode
            # 2x2 output tiles start at even rows / columns and read input rows 2i-1 .. 2i+2: the tiles whose patch holds (py, px)
            tiles_y = [i for i in range(H // 2) if 2 * i - 1 <= py <= 2 * i + 2]
            tiles_x = [j for j in range(W // 2) if 2 * j - 1 <= px <= 2 * j + 2]
            allowed = torch.zeros((N, H, W), dtype=torch.bool, device=DEV)
            for i in tiles_y:
                for j in tiles_x:
                    allowed[1, 2 * i:2 * i + 2, 2 * j:2 * j + 2] = True
            assert not bool((nf.any(dim=-1) & ~allowed).any()), "%s: non-finite values outside the tiles that read the pixel" % mode
            keep = ~allowed
            assert torch.equal(dirty[keep], clean[keep]), "%s: outputs of untouched tiles changed" % mode
            sets[mode] = nf.any(dim=-1)
        assert torch.equal(sets["f32"], sets["f32x3"])
    finally:
        engine.set_compute(prev)
