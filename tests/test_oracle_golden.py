"""Pin the CPU oracle (oracle/) against golden vectors produced by the reference itself
(tests/golden/make_goldens.py).  CPU only.  Tolerances: same torch build, same ops, so the float
results agree to rounding (rtol 1e-5 / atol 1e-6 unless stated); integer results are exact."""
import numpy as np
import pytest
import torch

from oracle import detgen, geometry, image_ops, kitti_gt, losses, nets

torch.set_num_threads(8)


def _close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def _check_summary(t, g, prefix, stride=97, rtol=1e-5, atol=1e-6):
    s = detgen.summarize(t, stride)
    assert tuple(s["shape"]) == tuple(g[prefix + "shape"])
    np.testing.assert_allclose(s["samples"], g[prefix + "samples"], rtol=rtol, atol=atol)
    np.testing.assert_allclose(s["sum"], g[prefix + "sum"], rtol=rtol, atol=atol * t.numel())
    np.testing.assert_allclose(s["abssum"], g[prefix + "abssum"], rtol=rtol, atol=atol * t.numel())
    np.testing.assert_allclose(s["max"], g[prefix + "max"], rtol=rtol, atol=atol)


def _params(sd):
    out = {}
    for k, v in sd.items():
        if torch.is_floating_point(v) and "running" not in k:
            v.requires_grad_(True)
        out[k] = v
    return out


def test_dispnets_config1(golden):
    g = golden("dispnets_cfg1")
    sd = _params(detgen.fill_state_dict(nets.dispnets_state_dict(), "dispnets"))
    x = detgen.image_batch(2, 128, 416, "dispnets:x")
    outs = nets.dispnets(sd, x, training=True)
    for i, o in enumerate(outs):
        _check_summary(o, g, "train%d_" % i)
    _close(outs[3], g["train3_full"])
    sum((o * detgen.uniform(tuple(o.shape), "dispnets:g%d" % i, -1, 1)).sum() for i, o in enumerate(outs)).backward()
    for key in ("conv1.0.weight", "conv7.2.weight", "upconv7.0.weight", "iconv3.0.weight", "predict_disp1.0.weight",
                "iconv1.0.bias"):
        _check_summary(sd[key].grad, g, "grad:%s:" % key, stride=53, rtol=1e-4, atol=1e-5)
    with torch.no_grad():
        _check_summary(nets.dispnets(sd, x, training=False), g, "eval_")


@pytest.mark.parametrize("tag,shape,full", [("vggbn_tiny", (2, 64, 96), True), ("vggbn_cfg", (2, 128, 416), False)])
def test_disp_vgg_bn(golden, tag, shape, full):
    g = golden(tag)
    b, h, w = shape
    sd = _params(detgen.fill_state_dict(nets.disp_vgg_bn_state_dict(), "vggbn"))
    x = detgen.image_batch(b, h, w, tag + ":x")
    gt = detgen.sparse_depth(b, h, w, tag + ":gt", density=0.3 if full else 0.05)
    disps = nets.disp_vgg_bn(sd, x, training=True)
    depth = [1 / d for d in disps]
    loss = losses.l1_loss(gt, depth, "kitti") + 0.1 * losses.smooth_loss(depth)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    for i, o in enumerate(disps):
        _check_summary(o, g, "disp%d_" % i, rtol=1e-4, atol=1e-5)
        if full:
            _close(o, g["disp%d" % i], rtol=1e-4, atol=1e-5)
    for key in [k[5:-6] for k in g.files if k.startswith("grad:") and k.endswith(":shape")]:
        _check_summary(sd[key].grad, g, "grad:%s:" % key, stride=53, rtol=2e-3, atol=1e-6)
    for key in [k[3:] for k in g.files if k.startswith("bn:features")]:
        _close(sd[key], g["bn:" + key], rtol=1e-4, atol=1e-6)
    assert int(sd["features.features.1.num_batches_tracked"]) == int(g["bn:nbt"])
    if full:
        with torch.no_grad():
            _close(nets.disp_vgg_bn(sd, x, training=False), g["eval_disp0"], rtol=1e-4, atol=1e-5)


def test_state_dict_keys_match_reference_layout():
    sd = nets.disp_vgg_bn_state_dict(with_classifier=False)
    assert len(sd) + 6 == 125                       # 125 keys incl. 6 classifier tensors
    assert "features.features.41.num_batches_tracked" in sd and "upconv4.0.weight" in sd
    assert tuple(sd["upconv4.0.weight"].shape) == (512, 256, 4, 4)
    assert tuple(sd["iconv2.0.weight"].shape) == (64, 193, 3, 3)
    n_used = sum(v.numel() for k, v in sd.items() if torch.is_floating_point(v) and "running" not in k)
    assert n_used == 19873156                       # params receiving grads
    assert sum(v.numel() for v in nets.dispnets_state_dict().values()) == 31596900


def test_supervised_losses(golden):
    g = golden("losses")
    b, h, w = 3, 32, 64
    gt = detgen.sparse_depth(b, h, w, "loss:gt", density=0.4, lo=0.5, hi=90.0)
    mk = lambda lo=1e-4, hi=95.0: [detgen.uniform((b, 1, h >> i, w >> i), "loss:pred%d" % i, lo, hi).requires_grad_()
                                   for i in range(4)]
    for ds in ("kitti", "nyu"):
        for name in ("l1_loss", "l2_loss", "berhu_loss", "Scale_invariant_loss"):
            if name == "berhu_loss" and ds == "nyu":
                continue
            depth = mk()
            v = getattr(losses, name)(gt, depth, ds)
            v.backward()
            np.testing.assert_allclose(v.item(), g["%s:%s" % (name, ds)], rtol=1e-5)
            _close(depth[0].grad, g["%s:%s:grad" % (name, ds)], rtol=1e-4, atol=1e-8)
    for key in [k for k in g.files if k.startswith("Multiscale") and ":grad" not in k]:
        name, _, pool = key.partition(":")
        depth = mk()
        v = getattr(losses, name)(gt, depth, pool) if pool else getattr(losses, name)(gt, depth)
        v.backward()
        np.testing.assert_allclose(v.item(), g[key], rtol=1e-5)
        for i in range(4):
            _close(depth[i].grad, g["%s:grad%d" % (key, i)], rtol=1e-4, atol=1e-8)
    depth = mk(0.5, 60.0)
    v = losses.smooth_loss(depth); v.backward()
    np.testing.assert_allclose(v.item(), g["smooth_loss"], rtol=1e-5)
    for i in range(4):
        _close(depth[i].grad, g["smooth_loss:grad%d" % i], rtol=1e-4, atol=1e-8)
    p = detgen.uniform((b, 8, h, w), "loss:ord", 0.0, 1.0).requires_grad_()
    v = losses.smooth_DORN_loss(p); v.backward()
    np.testing.assert_allclose(v.item(), g["smooth_DORN_loss"], rtol=1e-5)
    _close(p.grad, g["smooth_DORN_loss:grad"], rtol=1e-4, atol=1e-9)
    gt0 = gt.clone(); gt0[1] = 0
    v = losses.l1_loss(gt0, [detgen.uniform((b, 1, h, w), "loss:pred0", 1e-4, 95.0)], "kitti")
    assert np.isnan(g["l1_loss:empty_sample"]) and torch.isnan(v)
    m = [detgen.uniform((b, 2, h >> i, w >> i), "loss:mask%d" % i, 0.05, 0.95).requires_grad_() for i in range(2)]
    v = losses.explainability_loss(m); v.backward()
    np.testing.assert_allclose(v.item(), g["explainability_loss"], rtol=1e-5)
    _close(m[0].grad, g["explainability_loss:grad0"], rtol=1e-4, atol=1e-9)


def test_compute_errors(golden):
    g = golden("compute_errors")
    for ds, (b, h, w), hi in (("kitti", (3, 128, 416), 90.0), ("nyu", (2, 48, 64), 11.0)):
        gt = detgen.sparse_depth(b, h, w, "err:gt:" + ds, density=0.3, lo=0.5, hi=hi)
        pred = detgen.uniform((b, h, w), "err:pred:" + ds, 1e-4, hi)
        np.testing.assert_allclose(losses.compute_errors(gt, pred, ds), g["errors:%s" % ds], rtol=1e-6)
        np.testing.assert_allclose(losses.compute_errors(gt, pred, ds, True, True), g["errors:%s:median" % ds], rtol=1e-6)
    assert losses.garg_crop_bounds(128, 416) == (52, 126, 14, 401)


def _warp_inputs(b, h, w, tag):
    img = detgen.uniform((b, 3, h, w), tag + ":img", -1, 1)
    depth = detgen.uniform((b, h, w), tag + ":depth", 2.0, 30.0)
    pose = detgen.uniform((b, 6), tag + ":pose", -0.05, 0.05)
    fx, fy, cx, cy = 241.67 * w / 416, 246.28 * h / 128, 204.17 * w / 416, 59.0 * h / 128
    k = torch.tensor([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=torch.float32).repeat(b, 1, 1)
    return img, depth, pose, k


def test_inverse_warp_all_modes(golden):
    g = golden("warp")
    img, depth, pose, k = _warp_inputs(2, 24, 40, "warp")
    kinv = torch.from_numpy(g["kinv"])
    for ac in (False, True):
        for pad in ("zeros", "border"):
            for rot in ("euler", "quat"):
                d = depth.clone().requires_grad_(); p = pose.clone().requires_grad_()
                out = geometry.inverse_warp(img, d, p, k, kinv, rot, pad, align_corners=ac)
                (out * detgen.uniform(tuple(out.shape), "warp:g", -1, 1)).sum().backward()
                key = "ac%d:%s:%s" % (int(ac), pad, rot)
                _close(out, g[key], rtol=1e-5, atol=1e-6)
                _close(d.grad, g[key + ":gdepth"], rtol=1e-4, atol=1e-6)
                _close(p.grad, g[key + ":gpose"], rtol=1e-4, atol=1e-4)


def test_photometric_loss(golden):
    g = golden("warp")
    b, h, w = 2, 32, 64
    tgt, _, _, k = _warp_inputs(b, h, w, "photo")
    kinv = torch.inverse(k)
    refs = [detgen.uniform((b, 3, h, w), "photo:ref%d" % i, -1, 1) for i in range(2)]
    pose = detgen.uniform((b, 2, 6), "photo:pose", -0.03, 0.03)
    for ac in (False, True):
        for with_mask in (False, True):
            depth = [detgen.uniform((b, 1, h >> i, w >> i), "photo:d%d" % i, 2.0, 30.0).requires_grad_() for i in range(4)]
            pz = pose.clone().requires_grad_()
            mask = [detgen.uniform((b, 2, h >> i, w >> i), "photo:m%d" % i, 0.1, 0.9) for i in range(4)] if with_mask else [None] * 4
            v = losses.photometric_reconstruction_loss(tgt, refs, k, kinv, depth, mask, pz, "euler", "zeros", align_corners=ac)
            v.backward()
            key = "photo:ac%d:mask%d" % (int(ac), int(with_mask))
            np.testing.assert_allclose(v.item(), g[key], rtol=1e-5)
            _close(pz.grad, g[key + ":gpose"], rtol=1e-3, atol=1e-4)
            for i in range(4):
                _close(depth[i].grad, g[key + ":gdepth%d" % i], rtol=1e-4, atol=1e-7)
    depth = [detgen.uniform((b, 1, h >> i, w >> i), "photo:d%d" % i, 2.0, 30.0) for i in range(4)]
    v = losses.photometric_reconstruction_loss(tgt, refs, k, kinv, depth, None, pose, "euler", "zeros")
    np.testing.assert_allclose(v.item(), g["photo:bare_none_mask"], rtol=1e-5)


def test_ssim_and_edge_smoothness(golden):
    g = golden("layers")
    x = detgen.uniform((2, 3, 20, 28), "ssim:x", 0, 1).requires_grad_()
    y = detgen.uniform((2, 3, 20, 28), "ssim:y", 0, 1).requires_grad_()
    s = image_ops.ssim(x, y)
    (s * detgen.uniform(tuple(s.shape), "ssim:g", -1, 1)).sum().backward()
    _close(s, g["ssim"]); _close(x.grad, g["ssim_gx"], rtol=1e-4, atol=1e-6); _close(y.grad, g["ssim_gy"], rtol=1e-4, atol=1e-6)
    disp = detgen.uniform((2, 1, 20, 28), "esm:disp", 0.1, 5).requires_grad_()
    img = detgen.uniform((2, 3, 20, 28), "esm:img", 0, 1)
    e = image_ops.get_smooth_loss(disp, img); e.backward()
    np.testing.assert_allclose(e.item(), g["edge_smooth"], rtol=1e-6)
    _close(disp.grad, g["edge_smooth_gdisp"], rtol=1e-5, atol=1e-9)


def test_sid_and_ordinal_exact(golden):
    g = golden("dorn")
    for ds, hi in (("kitti", 85.0), ("nyu", 11.0)):
        d = detgen.uniform((2, 16, 24), "sid:d:" + ds, 0.0, hi)
        for kc in (71, 80):
            lab = image_ops.get_labels_sid(d, ordinal_c=kc, dataset=ds)
            assert lab.dtype == torch.int32
            np.testing.assert_array_equal(lab.numpy(), g["labels:%s:%d" % (ds, kc)])          # bit-exact
            np.testing.assert_array_equal(image_ops.get_depth_sid(lab, kc, ds).numpy(), g["decode:%s:%d" % (ds, kc)])
    pre = detgen.uniform((2, 24, 10, 14), "orl:pre", -3, 3).requires_grad_()
    dec, ordc = nets.ordinal_regression(pre)
    assert dec.dtype == torch.int64
    np.testing.assert_array_equal(dec.numpy(), g["orl:decode"])                               # bit-exact
    _close(ordc, g["orl:ord"], rtol=1e-6, atol=1e-7)
    gt = detgen.sparse_depth(2, 10, 14, "orl:gt", density=0.6, lo=0.5, hi=90)
    tgt = image_ops.get_labels_sid(gt, ordinal_c=12, dataset="kitti")
    v = losses.DORN_loss(gt, ordc, tgt, "kitti"); v.backward()
    np.testing.assert_allclose(v.item(), g["dorn_loss"], rtol=1e-5)
    _close(pre.grad, g["dorn_loss:gpre"], rtol=1e-4, atol=1e-8)
    sd = detgen.fill_state_dict(nets.disp_vgg_bn_state_dict(dorn_ordinal_c=8), "vggdorn")
    with torch.no_grad():
        dec, ordc = nets.disp_vgg_bn_dorn(sd, detgen.image_batch(1, 64, 96, "vggdorn:x"), training=False)
    _close(ordc, g["net:ord"], rtol=1e-4, atol=1e-5)
    assert (dec.numpy() != g["net:decode"]).mean() < 1e-3      # only P within rounding of 0.5 may flip


def test_kitti_ground_truth_exact(golden):
    import importlib.util, pathlib
    g = golden("kitti_gt")
    spec = importlib.util.spec_from_file_location("mk", pathlib.Path(__file__).parent / "golden" / "make_goldens.py")
    mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
    p_rect, r_rect, r, t, velo = mk.synthetic_kitti_scene()
    # the reference reads the calibration back from "%.6e" text; do the same rounding
    rt = lambda a: np.array([float("%.6e" % v) for v in a])
    velo2cam = np.hstack((rt(r).reshape(3, 3), rt(t)[..., np.newaxis]))
    velo = velo.copy(); velo[:, 3] = 1
    for shape in ((375, 1242), (120, 400)):
        depth = kitti_gt.generate_depth_map(velo, rt(p_rect), rt(r_rect), velo2cam, shape)
        mask = kitti_gt.generate_mask(depth, 1e-3, 80)
        yy, xx = np.nonzero(depth)
        np.testing.assert_array_equal(np.stack([yy, xx], 1).astype(np.int32), g["depth:%dx%d:yx" % shape])
        np.testing.assert_array_equal(depth[yy, xx], g["depth:%dx%d:val" % shape])            # bit-exact float64
        assert int(mask.sum()) == int(g["mask:%dx%d:count" % shape])
        np.testing.assert_array_equal(mask.sum(1).astype(np.int32), g["mask:%dx%d:rowsum" % shape])
    np.testing.assert_array_equal(np.array([kitti_gt.sub2ind((375, 1242), 10.0, 7.0), kitti_gt.sub2ind((120, 400), 119.0, 399.0)]),
                                  g["sub2ind"])
    assert list(kitti_gt.garg_crop(375, 1242)) == [153, 371, 44, 1197]


def test_train_step_golden(golden):
    """train.py:441-522 sequence: forward -> 1/disp -> l1 -> zero_grad/backward/Adam step, two iterations."""
    g = golden("trainstep_vggbn_l1")
    sd = _params(detgen.fill_state_dict(nets.disp_vgg_bn_state_dict(), "vggbn"))
    x = detgen.image_batch(2, 64, 96, "trainstep:x")
    gt = detgen.sparse_depth(2, 64, 96, "trainstep:gt", density=0.3)
    params = [v for k, v in sd.items() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.999))
    ls = []
    for _ in range(2):
        depth = [1 / d for d in nets.disp_vgg_bn(sd, x, training=True)]
        loss = losses.l1_loss(gt, depth, "kitti")
        opt.zero_grad(); loss.backward(); opt.step()
        ls.append(loss.item())
    np.testing.assert_allclose(ls, g["losses"], rtol=2e-5)
    for key in [k[5:-6] for k in g.files if k.startswith("post:") and k.endswith(":shape")]:
        _check_summary(sd[key], g, "post:%s:" % key, stride=31, rtol=1e-3, atol=2e-5)


# ------------------------------------------------------------------ ResNet / monodepth2-style nets / PoseExpNet (oracle/nets_res.py)
def _fresh_sd(product_module, prefix):
    """State dict with the reference's keys (taken from the product's parameter containers, whose layout is checked against the
    reference's key list below) filled with the closed-form values the goldens were generated with."""
    sd = {k: v.clone() for k, v in product_module.state_dict().items()}
    detgen.fill_state_dict(sd, prefix)
    return sd


def _grad_check(sd, g, prefix="grad:", rtol=2e-4):
    for key in sorted({k[len(prefix):].rsplit(":", 1)[0] for k in g.files if k.startswith(prefix)}):
        if (prefix + key + ":none") in g.files:
            assert sd[key].grad is None, key
            continue
        s = detgen.summarize(sd[key].grad, stride=53)
        scale = float(np.abs(g[prefix + key + ":samples"]).max()) + 1e-30
        np.testing.assert_allclose(s["samples"], g[prefix + key + ":samples"], rtol=rtol, atol=2e-5 * scale, err_msg=key)


def test_disp_res_50(golden):
    import supervised_dispnet_amd.models as models
    from oracle import nets_res
    g = golden("res50")
    b, h, w = 2, 64, 96
    sd = _params(_fresh_sd(models.Disp_res_50("nyu"), "res50"))
    x = detgen.image_batch(b, h, w, "res50:x")
    gt = detgen.sparse_depth(b, h, w, "res50:gt", density=0.6, lo=0.3, hi=11.0)
    disps = nets_res.disp_res_50(sd, x, training=True, datasets="nyu")
    depth = [1 / d for d in disps]
    loss = losses.l1_loss(gt, depth, "nyu") + 0.1 * losses.smooth_loss(depth)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    for i, d in enumerate(disps):
        _close(d, g["disp%d" % i], rtol=1e-4, atol=1e-5)
    _grad_check(sd, g)
    assert sd["bn1.weight"].grad is None                      # bn1's output is discarded by the reference (:141-145)
    for key in ("bn1.running_mean", "bn1.running_var", "layer4.2.bn3.running_mean", "layer1.0.downsample.1.running_var"):
        _close(sd[key], g["bn:" + key], rtol=1e-4, atol=1e-6)
    assert int(sd["bn1.num_batches_tracked"]) == int(g["bn1.num_batches_tracked"]) == 1
    with torch.no_grad():
        _close(nets_res.disp_res_50(sd, x, training=False, datasets="nyu"), g["eval_disp1"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("tag", ["res18", "res6", "res101", "vgg", "vggfeat"])
def test_model_zoo(golden, tag):
    """Disp_res_18 / Disp_res / Disp_res_101 / Disp_vgg / Disp_vgg_feature -- the oracle restatement against the
    imported reference's own outputs, gradients, BatchNorm buffers, eval output; and the product's state_dict layout against the
    reference's key list."""
    import supervised_dispnet_amd.models as models
    from tests.cases import zoo_cases
    g = golden("zoo")
    _, cls, kwargs, ds, run = [c for c in zoo_cases() if c[0] == tag][0]
    net = getattr(models, cls)(**kwargs)
    strip = lambda ks: sorted(k for k in ks if ".classifier." not in k)
    assert strip(net.state_dict().keys()) == strip(g[tag + ":keys"].tolist())        # drop-in checkpoint layout
    sd = _params(_fresh_sd(net, "zoo:" + tag))
    b, h, w = 2, 64, 96
    x = detgen.image_batch(b, h, w, "zoo:%s:x" % tag)
    gt = detgen.sparse_depth(b, h, w, "zoo:%s:gt" % tag, density=0.6, lo=0.3, hi=11.0)
    disps = run(sd, x, True)
    depth = [1 / d for d in disps]
    loss = losses.l1_loss(gt, depth, ds) + 0.1 * losses.smooth_loss(depth)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g[tag + ":loss"], rtol=1e-5)
    for i, d in enumerate(disps):
        _close(d, g["%s:disp%d" % (tag, i)], rtol=1e-4, atol=1e-5)
    _grad_check(sd, g, prefix=tag + ":grad:")
    for key in [k[len(tag) + 4:] for k in g.files if k.startswith(tag + ":bn:")]:
        _close(sd[key], g["%s:bn:%s" % (tag, key)], rtol=1e-4, atol=1e-6)
    with torch.no_grad():
        _close(run(sd, x, False), g[tag + ":eval"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("tag", ["fcrn", "res50_aspp", "deeplab"])
def test_model_zoo_fcrn_aspp(golden, tag):
    """SURVEY 8 f-4 tail: FCRN (up-projection blocks), res50_aspp / deeplab_depth (dilated bottlenecks, ceil-mode max-pool, summed ASPP
    classifier, align_corners resize) -- the oracle restatement against the imported reference's own output, gradients, BatchNorm
    buffers, eval output; the product's state_dict layout and its set of trainable parameters against the reference's."""
    import supervised_dispnet_amd.models as models
    from tests.cases import zoo2_cases, zoo2_dropout_mask
    g = golden("zoo2")
    _, cls, kwargs, run = [c for c in zoo2_cases() if c[0] == tag][0]
    net = getattr(models, cls)(**kwargs)
    assert sorted(net.state_dict().keys()) == sorted(g[tag + ":keys"].tolist())        # drop-in checkpoint layout
    assert sorted(k for k, q in net.named_parameters() if q.requires_grad) == sorted(g[tag + ":trainable"].tolist())
    assert sorted(id(q) for q in net._hot_parameters()) == sorted(id(q) for q in net.parameters() if q.requires_grad)
    assert sorted(id(q) for q in net._grad_production_order()) == sorted(id(q) for q in net._hot_parameters())
    sd = _params(_fresh_sd(net, "zoo2:" + tag))
    trainable = set(g[tag + ":trainable"].tolist())
    for k, v in sd.items():                                    # the ASPP nets freeze every BatchNorm's affine pair (models/ASPP.py:61-63)
        if v.requires_grad and k not in trainable:
            v.requires_grad_(False)
    b, h, w = 2, 64, 96
    x = detgen.image_batch(b, h, w, "zoo2:%s:x" % tag)
    gt = detgen.sparse_depth(b, h, w, "zoo2:%s:gt" % tag, density=0.6, lo=0.3, hi=11.0)
    disps = run(sd, x, True, zoo2_dropout_mask(tag))
    depth = [1 / d for d in disps]
    loss = losses.l1_loss(gt, depth, "kitti") + 0.1 * losses.smooth_loss(depth)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g[tag + ":loss"], rtol=1e-5)
    _close(disps[0], g[tag + ":disp0"], rtol=1e-4, atol=1e-5)
    _grad_check(sd, g, prefix=tag + ":grad:")
    for key in [k[len(tag) + 4:] for k in g.files if k.startswith(tag + ":bn:")]:
        _close(sd[key], g["%s:bn:%s" % (tag, key)], rtol=1e-4, atol=1e-6)
    with torch.no_grad():
        _close(run(sd, x, False), g[tag + ":eval"], rtol=1e-4, atol=1e-5)


def test_monodepth2_style_nets(golden):
    import supervised_dispnet_amd.models as models
    import supervised_dispnet_amd.networks as networks
    from oracle import nets_res
    g = golden("mono2")
    x = (detgen.image_batch(2, 64, 96, "mono2:x") + 1) / 2
    for tag, enc, which in (("vgg", networks.vggEncoder(16, False), "vgg"), ("res18", networks.ResnetEncoder(18, False), "18")):
        net = models.monodepth2(enc, networks.DepthDecoder(enc.num_ch_enc))
        keys = sorted(k for k in net.state_dict().keys() if ".classifier." not in k)
        want = sorted(k for k in g[tag + ":keys"].tolist() if ".classifier." not in k)
        assert keys == want                                    # drop-in checkpoint layout
        sd = _params(_fresh_sd(net, "mono2:" + tag))
        outs = nets_res.monodepth2(sd, x, which, training=True)
        ws = [detgen.uniform(tuple(o.shape), "mono2:g%d" % i, -1, 1) for i, o in enumerate(outs)]
        sum((o * wt).sum() for o, wt in zip(outs, ws)).backward()
        for i, o in enumerate(outs):
            _close(o, g["%s:disp%d" % (tag, i)], rtol=1e-4, atol=1e-5)
        _grad_check(sd, g, prefix=tag + ":grad:")
        with torch.no_grad():
            _close(nets_res.monodepth2(sd, x, which, training=False), g[tag + ":eval_disp0"], rtol=1e-4, atol=1e-5)


def test_pose_exp_net(golden):
    import supervised_dispnet_amd.models as models
    from oracle import nets_res
    g = golden("posenet")
    b, h, w = 2, 128, 416
    tgt = detgen.image_batch(b, h, w, "pose:tgt")
    refs = [detgen.image_batch(b, h, w, "pose:ref%d" % i) for i in range(2)]
    for exp in (False, True):
        tag = "exp%d" % int(exp)
        sd = _params(_fresh_sd(models.PoseExpNet(2, exp), "posenet"))
        masks, pose = nets_res.pose_exp_net(sd, tgt, refs, exp, training=True)
        _close(pose, g[tag + ":pose"], rtol=1e-4, atol=1e-7)
        loss = (pose * detgen.uniform(tuple(pose.shape), "pose:gp", -1, 1)).sum()
        if exp:
            for i, m in enumerate(masks):
                _check_summary(m, g, "%s:mask%d:" % (tag, i), rtol=1e-4, atol=1e-6)
                loss = loss + (m * detgen.uniform(tuple(m.shape), "pose:gm%d" % i, -1, 1)).sum()
        else:
            assert masks == [None] * 4
        loss.backward()
        _grad_check(sd, g, prefix=tag + ":grad:")
        with torch.no_grad():
            m1, pe = nets_res.pose_exp_net(sd, tgt, refs, exp, training=False)
        _close(pe, g[tag + ":eval_pose"], rtol=1e-4, atol=1e-7)
        if exp:
            np.testing.assert_allclose(detgen.summarize(m1)["samples"], g[tag + ":eval_mask1_samples"], rtol=1e-4, atol=1e-6)


def test_config3_pipeline_at_config_size(golden):
    """BASELINE configs[2] at 2 x 128 x 416: PoseExpNet -> Disp_vgg_BN -> 1/disp -> photometric + 0.1 * smooth, both nets'
    gradients (reference train.py:426-488, loss_functions.py:317-386)."""
    import supervised_dispnet_amd.models as models
    from oracle import nets_res
    from cases import config3_inputs
    g = golden("config3_cfg")
    tgt, refs, k, kinv = config3_inputs()
    _close(kinv, g["kinv"], rtol=1e-6, atol=1e-9)
    dsd = _params(detgen.fill_state_dict(nets.disp_vgg_bn_state_dict(), "vggbn"))
    psd = _params(_fresh_sd(models.PoseExpNet(2, False), "posenet"))
    mask, pose = nets_res.pose_exp_net(psd, tgt, refs, False, training=True)
    _close(pose, g["pose"], rtol=1e-4, atol=1e-7)
    disps = nets.disp_vgg_bn(dsd, tgt, training=True)
    depth = [1 / d for d in disps]
    l1 = losses.photometric_reconstruction_loss(tgt, refs, k, kinv, depth, mask, pose, "euler", "zeros")
    l3 = losses.smooth_loss(depth)
    (l1 + 0.1 * l3).backward()
    np.testing.assert_allclose(l1.item(), g["photo"], rtol=1e-5)
    np.testing.assert_allclose(l3.item(), g["smooth"], rtol=1e-5)
    for i, d in enumerate(disps):
        _check_summary(d, g, "disp%d_" % i)
    _grad_check(dsd, g, prefix="disp:grad:")
    _grad_check(psd, g, prefix="pose:grad:")
    _close(psd["pose_pred.bias"].grad, g["full:pose:grad:pose_pred.bias"], rtol=2e-4, atol=1e-7)


def test_dorn_ordinal_c_80_at_config_size(golden):
    """BASELINE configs[4] at 2 x 128 x 416 with ordinal_c = 80 (train.py:35): trunk -> Dropout2d (injected pattern) -> 1x1 conv
    to 160 logits -> OrdinalRegressionLayer -> DORN_loss, and the gradients of the head."""
    from cases import dorn80_inputs
    g = golden("dorn80_cfg")
    x, gt, mask = dorn80_inputs()
    sd = _params(detgen.fill_state_dict(nets.disp_vgg_bn_state_dict(dorn_ordinal_c=80), "vggdorn80"))
    tgt = image_ops.get_labels_sid(gt, ordinal_c=80, dataset="kitti")
    assert int(tgt.long().sum()) == int(g["labels_sum"])
    dec, ordc = nets.disp_vgg_bn_dorn(sd, x, training=True, dropout_mask=mask.view(-1, 16, 1, 1))
    loss = losses.DORN_loss(gt, ordc, tgt, "kitti")
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    _check_summary(ordc, g, "ord_")
    assert int(dec.sum()) == int(g["decode_sum"])
    np.testing.assert_array_equal(dec.reshape(-1)[::997].numpy(), g["decode_samples"])
    scale = float(np.abs(g["full:grad:conv_ord.weight"]).max())
    np.testing.assert_allclose(sd["conv_ord.weight"].grad.numpy(), g["full:grad:conv_ord.weight"], rtol=2e-4, atol=2e-5 * scale)
    np.testing.assert_allclose(sd["conv_ord.bias"].grad.numpy(), g["full:grad:conv_ord.bias"], rtol=2e-4, atol=2e-5 * float(np.abs(g["full:grad:conv_ord.bias"]).max()))
    _grad_check(sd, g)
