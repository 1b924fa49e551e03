"""GPU parity of the loss / geometry / ordinal kernels (HIP path through the C ABI) against the golden vectors captured from
the reference itself (tests/golden/*.npz, same closed-form inputs as tests/test_oracle_golden.py) and against the CPU
oracle on larger, ragged shapes.  pytest -m gpu.

Stated tolerances: scalar losses rtol 1e-5 (2e-5 where an exp/log chain is involved); per-pixel gradients rtol 1e-4 with an
absolute floor relative to max|ref|; warped images atol 2e-5 (a bilinear gather of fp32 coordinates: coordinate round-off
of ~1e-6 px times image gradient); pose gradients rtol 1e-3 (sums over all pixels); integer outputs exact.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import supervised_dispnet_amd.inverse_warp as IW  # noqa: E402
import supervised_dispnet_amd.layers as L  # noqa: E402
import supervised_dispnet_amd.loss_functions as LF  # noqa: E402
import supervised_dispnet_amd.utils as U  # noqa: E402
from oracle import detgen, geometry as OG, image_ops as OI, losses as OL, nets as ON  # noqa: E402  (the checker)

DEV = torch.device("cuda:0")


def close(name, got, want, rtol, atol_rel=0.0, atol=0.0, max_bad=0):
    got = got.detach().double().cpu()
    want = torch.as_tensor(np.asarray(want)).double() if not torch.is_tensor(want) else want.detach().double().cpu()
    assert got.shape == want.shape, "%s: shape %s vs %s" % (name, tuple(got.shape), tuple(want.shape))
    scale = float(want.abs().max()) + 1e-30
    err = (got - want).abs()
    tol = atol + atol_rel * scale + rtol * want.abs()
    bad = err > tol
    if int(bad.sum()) > max_bad:
        idx = np.unravel_index(int(torch.argmax(err - tol)), got.shape)
        raise AssertionError("%s: %d/%d off; worst at %s got %.9g want %.9g (max|want| %.4g, max err %.4g)" % (
            name, int(bad.sum()), got.numel(), idx, float(got[idx]), float(want[idx]), scale, float(err.max())))


def dev(t):
    return t.to(DEV)


def test_supervised_losses_golden(golden):
    g = golden("losses")
    b, h, w = 3, 32, 64
    gt = detgen.sparse_depth(b, h, w, "loss:gt", density=0.4, lo=0.5, hi=90.0)
    mk = lambda lo=1e-4, hi=95.0: [dev(detgen.uniform((b, 1, h >> i, w >> i), "loss:pred%d" % i, lo, hi)).requires_grad_() for i in range(4)]
    for ds in ("kitti", "nyu"):
        for name in ("l1_loss", "l2_loss", "berhu_loss", "Scale_invariant_loss"):
            if name == "berhu_loss" and ds == "nyu":
                continue
            depth = mk()
            v = getattr(LF, name)(dev(gt), depth, ds)
            v.backward()
            np.testing.assert_allclose(v.item(), g["%s:%s" % (name, ds)], rtol=1e-5, err_msg=name + ds)
            close("%s:%s:grad" % (name, ds), depth[0].grad, g["%s:%s:grad" % (name, ds)], rtol=1e-4, atol_rel=1e-6)
    # berHu on nyu: the reference crashes (NameError); the oracle's documented position = loop like kitti with max depth 10
    depth = mk()
    v = LF.berhu_loss(dev(gt), depth, "nyu")
    v.backward()
    od = [detgen.uniform((b, 1, h, w), "loss:pred0", 1e-4, 95.0).requires_grad_()]
    ov = OL.berhu_loss(gt, od, "nyu")
    ov.backward()
    np.testing.assert_allclose(v.item(), ov.item(), rtol=1e-5)
    close("berhu:nyu:grad", depth[0].grad, od[0].grad, rtol=1e-4, atol_rel=1e-6)
    for key in [k for k in g.files if k.startswith("Multiscale") and ":grad" not in k]:
        name, _, pool = key.partition(":")
        depth = mk()
        v = getattr(LF, name)(dev(gt), depth, pool) if pool else getattr(LF, name)(dev(gt), depth)
        v.backward()
        np.testing.assert_allclose(v.item(), g[key], rtol=1e-5, err_msg=key)
        for i in range(4):
            close("%s:grad%d" % (key, i), depth[i].grad, g["%s:grad%d" % (key, i)], rtol=1e-4, atol_rel=1e-6)
    # bilinear F.upsample variant of Multiscale_FULL_L1_loss (the reference's default pool_type) vs the oracle
    depth = mk()
    v = LF.Multiscale_FULL_L1_loss(dev(gt), depth)
    v.backward()
    od = [detgen.uniform((b, 1, h >> i, w >> i), "loss:pred%d" % i, 1e-4, 95.0).requires_grad_() for i in range(4)]
    ov = OL.Multiscale_FULL_L1_loss(gt, od, "bilinear")
    ov.backward()
    np.testing.assert_allclose(v.item(), ov.item(), rtol=1e-5)
    for i in range(4):
        close("full_l1:bilinear:grad%d" % i, depth[i].grad, od[i].grad, rtol=1e-4, atol_rel=1e-6)
    gt0 = gt.clone()
    gt0[1] = 0
    v = LF.l1_loss(dev(gt0), [dev(detgen.uniform((b, 1, h, w), "loss:pred0", 1e-4, 95.0))], "kitti")
    assert torch.isnan(v).item() and np.isnan(g["l1_loss:empty_sample"])      # empty mask -> NaN like the reference
    m = [dev(detgen.uniform((b, 2, h >> i, w >> i), "loss:mask%d" % i, 0.05, 0.95)).requires_grad_() for i in range(2)]
    v = LF.explainability_loss(m)
    v.backward()
    np.testing.assert_allclose(v.item(), g["explainability_loss"], rtol=1e-5)
    close("explainability:grad0", m[0].grad, g["explainability_loss:grad0"], rtol=1e-4, atol_rel=1e-7)
    with pytest.raises(ValueError):
        LF.l1_loss(dev(gt), mk(), "cityscapes")


def test_pyramids_exact():
    """GT pyramids are value-selecting / short fp32 expressions with a fixed operation order: they must equal torch-CPU's bits."""
    gt = detgen.sparse_depth(3, 34, 66, "pyr:gt", density=0.5, lo=0.5, hi=90.0)     # ragged: 34 -> 17 -> 8 -> 4
    for name in ("max", "avg", "bilinear"):
        ours = getattr(LF, "generate_%s_pyramid" % name)(dev(gt))
        want = getattr(OL, "generate_%s_pyramid" % name)(gt)
        assert len(ours) == 4
        for a, b_ in zip(ours, want):
            np.testing.assert_array_equal(a.cpu().numpy(), b_.reshape(a.shape).numpy())


def test_per_sample_losses_ragged_vs_oracle():
    """Shapes that are not multiples of anything, dense and near-empty masks, big (multi-split) groups."""
    for (b, h, w, dens) in ((5, 37, 53, 0.7), (2, 128, 416, 0.05), (1, 200, 300, 1.0)):
        gt = detgen.sparse_depth(b, h, w, "rag:gt%d" % h, density=dens, lo=0.5, hi=85.0)
        for name in ("l1_loss", "l2_loss", "berhu_loss", "Scale_invariant_loss"):
            p = detgen.uniform((b, 1, h, w), "rag:p%d" % h, 1e-4, 90.0)
            d = dev(p).requires_grad_()
            v = getattr(LF, name)(dev(gt), [d], "kitti")
            v.backward()
            od = p.clone().requires_grad_()
            ov = getattr(OL, name)(gt, [od], "kitti")
            ov.backward()
            np.testing.assert_allclose(v.item(), ov.item(), rtol=2e-5, err_msg="%s %dx%d" % (name, h, w))
            close("%s:%dx%d:grad" % (name, h, w), d.grad, od.grad, rtol=2e-4, atol_rel=2e-6)


def test_fused_masked_forward_is_bitwise_the_three_launches():
    """dn_masked_loss_fwd_fused (one launch: the block that arrives last reduces the split partials and divides) against the three launches
    of dn_masked_loss_fwd: loss, statistics-dependent gradients and the Multiscale accumulation bit for bit -- per-sample groups incl. the
    metric's 32 x 128 x 416 (64 splits x 32 groups = 2048 blocks), one whole-batch group, a one-block case, an empty mask (NaN)."""
    cases = ((32, 128, 416, 0.05), (5, 37, 53, 0.7), (1, 8, 8, 1.0), (3, 64, 96, 0.3))
    try:
        for (b, h, w, dens) in cases:
            gt = dev(detgen.sparse_depth(b, h, w, "fuse:gt%d" % h, density=dens, lo=0.5, hi=85.0))
            p = detgen.uniform((b, 1, h, w), "fuse:p%d" % h, 1e-4, 90.0)
            for name in ("l1_loss", "l2_loss", "Scale_invariant_loss"):
                out = {}
                for fused in (True, False, True):          # (the second fused run reuses the self-resetting counters)
                    LF.FUSE_MASKED_FWD = fused
                    d = dev(p).requires_grad_()
                    v = getattr(LF, name)(gt, [d], "kitti")
                    v.backward()
                    torch.cuda.synchronize()
                    if fused in out:
                        assert torch.equal(out[fused][0], v.detach()) and torch.equal(out[fused][1], d.grad)
                    out[fused] = (v.detach().clone(), d.grad.clone())
                assert torch.equal(out[True][0], out[False][0]), (name, b, h, w, float(out[True][0]), float(out[False][0]))
                assert torch.equal(out[True][1], out[False][1]), (name, b, h, w)
        gt = dev(detgen.sparse_depth(3, 32, 64, "fuse:ms", density=0.4, lo=0.5, hi=90.0))
        res = {}
        for fused in (True, False):
            LF.FUSE_MASKED_FWD = fused
            depth = [dev(detgen.uniform((3, 1, 32 >> i, 64 >> i), "fuse:msp%d" % i, 1e-4, 95.0)).requires_grad_() for i in range(4)]
            v = LF.Multiscale_L1_loss(gt, depth, "max")
            v.backward()
            res[fused] = [v.detach().clone()] + [x.grad.clone() for x in depth]
        for u, v in zip(res[True], res[False]):
            assert torch.equal(u, v)
        gt0 = gt.clone()
        gt0[1] = 0
        LF.FUSE_MASKED_FWD = True
        assert torch.isnan(LF.l1_loss(gt0, [dev(detgen.uniform((3, 1, 32, 64), "fuse:e", 1e-4, 95.0))], "kitti")).item()
        assert int(LF._FOLD_COUNTERS[DEV][0].abs().sum().item()) == 0           # every launch left its counter zero
    finally:
        LF.FUSE_MASKED_FWD = True


def test_compute_errors_golden_incl_median(golden):
    g = golden("compute_errors")
    for ds, (b, h, w), hi in (("kitti", (3, 128, 416), 90.0), ("nyu", (2, 48, 64), 11.0)):
        gt = detgen.sparse_depth(b, h, w, "err:gt:" + ds, density=0.3, lo=0.5, hi=hi)
        pred = detgen.uniform((b, h, w), "err:pred:" + ds, 1e-4, hi)
        np.testing.assert_allclose(LF.compute_errors(dev(gt), dev(pred), ds), g["errors:%s" % ds], rtol=1e-4)
        np.testing.assert_allclose(LF.compute_errors(dev(gt), dev(pred), ds, True, True), g["errors:%s:median" % ds], rtol=1e-4)


def _warp_inputs(b, h, w, tag):
    img = detgen.uniform((b, 3, h, w), tag + ":img", -1, 1)
    depth = detgen.uniform((b, h, w), tag + ":depth", 2.0, 30.0)
    pose = detgen.uniform((b, 6), tag + ":pose", -0.05, 0.05)
    fx, fy, cx, cy = 241.67 * w / 416, 246.28 * h / 128, 204.17 * w / 416, 59.0 * h / 128
    k = torch.tensor([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=torch.float32).repeat(b, 1, 1)
    return img, depth, pose, k


def test_inverse_warp_all_modes_golden(golden):
    g = golden("warp")
    img, depth, pose, k = _warp_inputs(2, 24, 40, "warp")
    kinv = torch.from_numpy(g["kinv"])
    gw = detgen.uniform((2, 3, 24, 40), "warp:g", -1, 1)
    for ac in (False, True):
        for pad in ("zeros", "border"):
            for rot in ("euler", "quat"):
                d = dev(depth).requires_grad_()
                p = dev(pose).requires_grad_()
                out = IW.inverse_warp(dev(img), d, p, dev(k), dev(kinv), rot, pad, align_corners=ac)
                (out * dev(gw)).sum().backward()
                key = "ac%d:%s:%s" % (int(ac), pad, rot)
                # a sample whose coordinate lands within round-off of a pixel boundary / the [-1,1] edge may pick the other
                # side: allow a handful of such pixels, everything else to 2e-5
                close(key, out, g[key], rtol=1e-4, atol=2e-5, max_bad=3)
                close(key + ":gdepth", d.grad, g[key + ":gdepth"], rtol=2e-3, atol_rel=2e-4, max_bad=3)
                close(key + ":gpose", p.grad, g[key + ":gpose"], rtol=2e-3, atol_rel=2e-3)
    with pytest.raises(AssertionError):
        IW.inverse_warp(dev(img), dev(depth)[:, None], dev(pose), dev(k), dev(kinv))        # check_sizes like the reference


def test_pose_vec2mat_matches_oracle():
    pose = detgen.uniform((5, 6), "pv2m", -0.7, 0.7)
    for rot in ("euler", "quat"):
        close("pose_vec2mat:" + rot, IW.pose_vec2mat(dev(pose), rot), OG.pose_vec2mat(pose, rot), rtol=1e-5, atol=1e-6)
    close("euler2mat", IW.euler2mat(dev(pose[:, 3:])), OG.euler2mat(pose[:, 3:]), rtol=1e-5, atol=1e-6)
    close("quat2mat", IW.quat2mat(dev(pose[:, 3:])), OG.quat2mat(pose[:, 3:]), rtol=1e-5, atol=1e-6)
    # differentiable like the reference's torch expressions (inverse_warp.py:77-157)
    for rot in ("euler", "quat"):
        wt = torch.randn(pose.shape[0], 3, 4, generator=torch.Generator().manual_seed(3))
        pc = pose.clone().requires_grad_(True)
        (OG.pose_vec2mat(pc, rot) * wt).sum().backward()
        pd = dev(pose).requires_grad_(True)
        (IW.pose_vec2mat(pd, rot) * dev(wt)).sum().backward()
        close("d pose_vec2mat:" + rot, pd.grad, pc.grad, rtol=1e-4, atol=1e-6)
        ac = pose[:, 3:].clone().requires_grad_(True)
        fn_o, fn_h = (OG.euler2mat, IW.euler2mat) if rot == "euler" else (OG.quat2mat, IW.quat2mat)
        (fn_o(ac) * wt[:, :, :3]).sum().backward()
        ad = dev(pose[:, 3:]).requires_grad_(True)
        (fn_h(ad) * dev(wt[:, :, :3])).sum().backward()
        close("d %s2mat" % rot, ad.grad, ac.grad, rtol=1e-4, atol=1e-6)


def test_photometric_loss_golden(golden):
    g = golden("warp")
    b, h, w = 2, 32, 64
    tgt, _, _, k = _warp_inputs(b, h, w, "photo")
    kinv = torch.inverse(k)
    refs = [detgen.uniform((b, 3, h, w), "photo:ref%d" % i, -1, 1) for i in range(2)]
    pose = detgen.uniform((b, 2, 6), "photo:pose", -0.03, 0.03)
    for ac in (False, True):
        for with_mask in (False, True):
            depth = [dev(detgen.uniform((b, 1, h >> i, w >> i), "photo:d%d" % i, 2.0, 30.0)).requires_grad_() for i in range(4)]
            pz = dev(pose).requires_grad_()
            mask = [dev(detgen.uniform((b, 2, h >> i, w >> i), "photo:m%d" % i, 0.1, 0.9)) for i in range(4)] if with_mask else [None] * 4
            v = LF.photometric_reconstruction_loss(dev(tgt), [dev(r) for r in refs], dev(k), dev(kinv), depth, mask, pz, "euler", "zeros",
                                                   align_corners=ac)
            v.backward()
            key = "photo:ac%d:mask%d" % (int(ac), int(with_mask))
            np.testing.assert_allclose(v.item(), g[key], rtol=2e-5, err_msg=key)
            close(key + ":gpose", pz.grad, g[key + ":gpose"], rtol=2e-3, atol_rel=2e-3)
            for i in range(4):
                close(key + ":gdepth%d" % i, depth[i].grad, g[key + ":gdepth%d" % i], rtol=2e-3, atol_rel=2e-4, max_bad=3)
    depth = [dev(detgen.uniform((b, 1, h >> i, w >> i), "photo:d%d" % i, 2.0, 30.0)) for i in range(4)]
    v = LF.photometric_reconstruction_loss(dev(tgt), [dev(r) for r in refs], dev(k), dev(kinv), depth, None, dev(pose), "euler", "zeros")
    np.testing.assert_allclose(v.item(), g["photo:bare_none_mask"], rtol=2e-5)


def test_photometric_mask_gradient_vs_oracle():
    b, h, w = 2, 32, 64
    tgt, _, _, k = _warp_inputs(b, h, w, "photo")
    kinv = torch.inverse(k)
    refs = [detgen.uniform((b, 3, h, w), "photo:ref%d" % i, -1, 1) for i in range(2)]
    pose = detgen.uniform((b, 2, 6), "photo:pose", -0.03, 0.03)
    depth = [detgen.uniform((b, 1, h >> i, w >> i), "photo:d%d" % i, 2.0, 30.0) for i in range(2)]
    masks = [detgen.uniform((b, 2, h >> i, w >> i), "photo:m%d" % i, 0.1, 0.9) for i in range(2)]
    om = [m.clone().requires_grad_() for m in masks]
    OL.photometric_reconstruction_loss(tgt, refs, k, kinv, depth, om, pose, "quat", "border").backward()
    gm = [dev(m).requires_grad_() for m in masks]
    v = LF.photometric_reconstruction_loss(dev(tgt), [dev(r) for r in refs], dev(k), dev(kinv), [dev(d) for d in depth], gm, dev(pose),
                                           "quat", "border")
    v.backward()
    for i in range(2):
        close("dmask%d" % i, gm[i].grad, om[i].grad, rtol=1e-3, atol_rel=1e-4, max_bad=3)


def test_ssim_and_edge_smoothness_golden(golden):
    g = golden("layers")
    x = dev(detgen.uniform((2, 3, 20, 28), "ssim:x", 0, 1)).requires_grad_()
    y = dev(detgen.uniform((2, 3, 20, 28), "ssim:y", 0, 1)).requires_grad_()
    s = L.SSIM()(x, y)
    (s * dev(detgen.uniform(tuple(s.shape), "ssim:g", -1, 1))).sum().backward()
    close("ssim", s, g["ssim"], rtol=1e-4, atol=1e-5)
    close("ssim_gx", x.grad, g["ssim_gx"], rtol=1e-3, atol_rel=1e-4)
    close("ssim_gy", y.grad, g["ssim_gy"], rtol=1e-3, atol_rel=1e-4)
    disp = dev(detgen.uniform((2, 1, 20, 28), "esm:disp", 0.1, 5)).requires_grad_()
    img = dev(detgen.uniform((2, 3, 20, 28), "esm:img", 0, 1))
    e = L.get_smooth_loss(disp, img)
    e.backward()
    np.testing.assert_allclose(e.item(), g["edge_smooth"], rtol=1e-5)
    close("edge_smooth_gdisp", disp.grad, g["edge_smooth_gdisp"], rtol=1e-4, atol_rel=1e-6)


def test_sid_and_ordinal_exact_golden(golden):
    g = golden("dorn")
    for ds, hi in (("kitti", 85.0), ("nyu", 11.0)):
        d = detgen.uniform((2, 16, 24), "sid:d:" + ds, 0.0, hi)
        for kc in (71, 80):
            lab = U.get_labels_sid(dev(d), ordinal_c=kc, dataset=ds)
            assert lab.dtype == torch.int32
            np.testing.assert_array_equal(lab.cpu().numpy(), g["labels:%s:%d" % (ds, kc)])          # bit-exact integers
            np.testing.assert_allclose(U.get_depth_sid(lab, kc, ds).cpu().numpy(), g["decode:%s:%d" % (ds, kc)], rtol=2e-6)
    # larger exactness sweep against the oracle (torch-CPU float32 expression)
    d = detgen.uniform((4, 128, 416), "sid:big", 0.0, 85.0)
    np.testing.assert_array_equal(U.get_labels_sid(dev(d), 80, "kitti").cpu().numpy(), OI.get_labels_sid(d, 80, "kitti").numpy())


def test_dorn_loss_golden(golden):
    from supervised_dispnet_amd.models.Disp_vgg_BN_DORN import OrdinalRegressionLayer
    g = golden("dorn")
    pre = dev(detgen.uniform((2, 24, 10, 14), "orl:pre", -3, 3)).requires_grad_()
    dec, ordc = OrdinalRegressionLayer()(pre)
    assert dec.dtype == torch.int64 and tuple(dec.shape) == (2, 1, 10, 14)
    np.testing.assert_array_equal(dec.cpu().numpy(), g["orl:decode"])                               # bit-exact
    close("orl:ord", ordc, g["orl:ord"], rtol=1e-5, atol=1e-7)
    gt = detgen.sparse_depth(2, 10, 14, "orl:gt", density=0.6, lo=0.5, hi=90)
    tgt = U.get_labels_sid(dev(gt), ordinal_c=12, dataset="kitti")
    v = LF.DORN_loss(dev(gt), ordc, tgt, "kitti")
    v.backward()
    np.testing.assert_allclose(v.item(), g["dorn_loss"], rtol=1e-5)
    close("dorn_loss:gpre", pre.grad, g["dorn_loss:gpre"], rtol=1e-4, atol_rel=1e-6)
