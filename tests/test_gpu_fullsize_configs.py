"""BASELINE configs[2], [3], [4] on the GPU at THEIR OWN sizes (round-1 only exercised them at toy size):

  config 3  Disp_vgg_BN + PoseExpNet + photometric_reconstruction_loss + smooth_loss, seq-len 3, 128x416:
            2 x 128 x 416 against the CPU oracle and the reference's golden (tests/golden/config3_cfg.npz);
            32 x 128 x 416 through batch-partition identities of the loss kernels and one whole training step.
  config 4  Disp_res_50 (+ monodepth2-style ResnetEncoder(50) + DepthDecoder) at 16 x 480 x 640: adjoint identities, linearity and
            a subsampled receptive-field check for every conv flavour the net uses at the shapes it uses them (7x7/2 stem, 1x1,
            1x1/2, 3x3/2, 3x3 on odd 15x20 maps, transposed 3x3/2 with output padding, virtual concat, reflection padding with a
            x2 nearest operand), the pointwise passes (3x3/2 max-pool, bottleneck tail) against PyTorch-CPU, one whole step.
  config 5  Disp_vgg_BN_DORN with ordinal_c = 80 at 128x416: b = 2 against the reference's golden, b = 4 against the oracle
            (ordinal probabilities, decode, loss, every gradient incl. conv_ord), b = 32 one whole step.
The CPU oracle needs minutes at the full batches, hence the size-independent properties there (as tests/test_gpu_fullsize.py).
"""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import supervised_dispnet_amd.loss_functions as LF  # noqa: E402
import supervised_dispnet_amd.models as models  # noqa: E402
import supervised_dispnet_amd.networks as networks  # noqa: E402
import supervised_dispnet_amd.utils as U  # noqa: E402
from cases import config3_inputs, dorn80_inputs  # noqa: E402
from oracle import detgen, image_ops as OI, losses as OL, nets as ON, nets_res  # noqa: E402  (the checker)
from supervised_dispnet_amd import engine  # noqa: E402
from supervised_dispnet_amd._lib import ACT_NONE  # noqa: E402
from supervised_dispnet_amd.functional import reciprocal  # noqa: E402
from supervised_dispnet_amd.optim import FusedAdam  # noqa: E402
from test_gpu_models import _check_all_grads, _fresh, _is_pre_bn_conv_bias, _oracle_params, close, grad_close  # noqa: E402

DEV = torch.device("cuda:0")


def dot64(a, b):
    return float((a.double() * b.double()).sum())


def _labels_match(got, want, depth, K, beta=80.999):
    """SID labels int(K * log(d + 0.999) / log(beta)) are integer results and must be exact -- except where the float32 expression
    lands within a few ulp of an integer: torch-CPU's vectorised logf (Sleef, <= 1 ulp) is not correctly rounded (and differs
    between its vector body and scalar tail), the HIP kernel evaluates the correctly rounded float32 expression, so the truncation
    may fall on either side there.  Everything else is compared exactly."""
    got, want = np.asarray(got).reshape(-1), np.asarray(want).reshape(-1)
    diff = np.nonzero(got != want)[0]
    if diff.size == 0:
        return
    v = K * np.log(np.asarray(depth, dtype=np.float64).reshape(-1)[diff] + 0.999) / np.log(beta)
    assert np.all(np.abs(got[diff] - want[diff]) == 1), "labels differ by more than one bin"
    assert np.all(np.abs(v - np.round(v)) < 6e-5), "labels differ away from a rounding boundary: %s" % v
    assert diff.size <= 1e-4 * got.size


def _summary_close(name, t, g, prefix, rtol=1e-3, atol_rel=1e-4):
    s = detgen.summarize(t.detach().float().cpu())
    want = g[prefix + "samples"]
    assert tuple(s["shape"]) == tuple(g[prefix + "shape"]), name
    np.testing.assert_allclose(s["samples"], want, rtol=rtol, atol=atol_rel * float(np.abs(want).max()), err_msg=name)


# ================================================================================================ config 3
def _config3_hip(b, tgt, refs, k, kinv, disp_net, pose_net):
    mask, pose = pose_net(tgt, refs)
    disps = disp_net(tgt)
    depth = [reciprocal(d) for d in disps]
    l1 = LF.photometric_reconstruction_loss(tgt, refs, k, kinv, depth, mask, pose, "euler", "zeros")
    l3 = LF.smooth_loss(depth)
    return mask, pose, disps, depth, l1, l3


def test_config3_matches_oracle_and_reference_golden_at_128x416(golden):
    g = golden("config3_cfg")
    tgt, refs, k, kinv = config3_inputs()
    disp_net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
    dsd0 = _fresh(disp_net, "vggbn")
    pose_net = models.PoseExpNet(nb_ref_imgs=2, output_exp=False)
    psd0 = _fresh(pose_net, "posenet")
    disp_net.to(DEV).train()
    pose_net.to(DEV).train()
    mask, pose, disps, depth, l1, l3 = _config3_hip(2, tgt.to(DEV), [r.to(DEV) for r in refs], k.to(DEV), kinv.to(DEV), disp_net, pose_net)
    (l1 + 0.1 * l3).backward()
    torch.cuda.synchronize()
    # the reference's own numbers
    np.testing.assert_allclose(l1.item(), float(g["photo"]), rtol=2e-4)
    np.testing.assert_allclose(l3.item(), float(g["smooth"]), rtol=2e-4)
    close("pose(golden)", pose, g["pose"], rtol=1e-3, atol_rel=1e-4)
    for i, d in enumerate(disps):
        _summary_close("disp%d(golden)" % i, d, g, "disp%d_" % i)
    # the oracle on the same inputs: every gradient of both nets
    dsd, psd = _oracle_params(dsd0), _oracle_params(psd0)
    omask, opose = nets_res.pose_exp_net(psd, tgt, refs, False, training=True)
    odisps = ON.disp_vgg_bn(dsd, tgt, training=True)
    odepth = [1 / d for d in odisps]
    (OL.photometric_reconstruction_loss(tgt, refs, k, kinv, odepth, omask, opose, "euler", "zeros") + 0.1 * OL.smooth_loss(odepth)).backward()
    for name, p in disp_net.named_parameters():
        if _is_pre_bn_conv_bias(name):
            continue
        grad_close("disp grad:" + name, p.grad, dsd[name].grad)
    _check_all_grads(pose_net, psd)
    close("pose_pred.bias grad (golden)", pose_net.pose_pred.bias.grad, g["full:pose:grad:pose_pred.bias"], rtol=1e-2, atol_rel=1e-3)


def test_config3_full_batch_32x128x416_properties():
    """The metric-size batch.  Each (scale, reference) term of the photometric loss is a mean over B*3*h*w, the smoothness terms are
    means over B*h*w: for a batch split into equal chunks the loss is the mean of the chunks' losses and the gradient w.r.t. a
    chunk's depth / pose is 1/nchunks of the chunk-only gradient -- exact identities that tie the full-size launch to the 8-image
    launches (whose kernels the 2 x 128 x 416 oracle test pins).  Then one whole training step of both nets."""
    B, H, W = 32, 128, 416
    g = torch.Generator().manual_seed(3)
    tgt = ((torch.rand(B, 3, H, W, generator=g) - 0.5) / 0.5).to(DEV)
    refs = [(tgt.cpu() + 0.05 * torch.randn(B, 3, H, W, generator=g)).clamp(-1, 1).to(DEV) for _ in range(2)]
    K = torch.tensor([[241.67, 0, 204.17], [0, 246.28, 59.0], [0, 0, 1]], dtype=torch.float32)
    k, kinv = K.repeat(B, 1, 1).to(DEV), torch.inverse(K).repeat(B, 1, 1).to(DEV)
    torch.manual_seed(0)
    disp_net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
    disp_net.init_weights(use_pretrained_weights=False)
    pose_net = models.PoseExpNet(nb_ref_imgs=2, output_exp=False)
    pose_net.init_weights()
    disp_net.to(DEV).train()
    pose_net.to(DEV).train()
    with torch.no_grad():
        _m, pose0 = pose_net(tgt, refs)
        depth0 = [reciprocal(d) for d in disp_net(tgt)]
    assert tuple(pose0.shape) == (B, 2, 6)

    def loss_and_grads(sl):
        dep = [d[sl].clone().requires_grad_() for d in depth0]
        po = pose0[sl].clone().requires_grad_()
        rr = [r[sl].contiguous() for r in refs]
        v = LF.photometric_reconstruction_loss(tgt[sl].contiguous(), rr, k[sl].contiguous(), kinv[sl].contiguous(), dep, [None] * 4, po,
                                               "euler", "zeros") + 0.1 * LF.smooth_loss(dep)
        v.backward()
        return float(v.item()), [d.grad for d in dep], po.grad

    full, gd_full, gp_full = loss_and_grads(slice(0, B))
    assert np.isfinite(full) and all(torch.isfinite(x).all() for x in gd_full) and torch.isfinite(gp_full).all()
    assert float(gp_full.abs().max()) > 0 and all(float(x.abs().max()) > 0 for x in gd_full)
    nch, acc = 4, 0.0
    for c in range(nch):
        sl = slice(c * 8, c * 8 + 8)
        v, gd, gp = loss_and_grads(sl)
        acc += v / nch
        for i in range(4):
            want = gd[i] / nch
            assert torch.allclose(gd_full[i][sl], want, rtol=1e-4, atol=1e-6 * float(want.abs().max())), "d(depth%d) chunk %d" % (i, c)
        assert torch.allclose(gp_full[sl], gp / nch, rtol=2e-3, atol=2e-5 * float(gp.abs().max() / nch)), "d(pose) chunk %d" % c
    np.testing.assert_allclose(full, acc, rtol=2e-5)
    # ---- one whole step of the config (both nets trained, as bench.py --config photo128 does)
    params = list(disp_net._hot_parameters()) + list(pose_net._hot_parameters())
    opt = FusedAdam(params, lr=1e-4, betas=(0.9, 0.999), production_order=disp_net._grad_production_order())
    before = opt.arena.flat_p.clone()
    mask, pose, disps, depth, l1, l3 = _config3_hip(B, tgt, refs, k, kinv, disp_net, pose_net)
    loss = l1 + 0.1 * l3
    opt.zero_grad()
    loss.backward()
    assert np.isfinite(loss.item()) and torch.isfinite(opt.arena.flat_g).all()
    live = sum(float(p._dn_grad_view.abs().max()) > 0.0 for p in opt.arena.params)
    assert live >= 0.75 * len(opt.arena.params)
    for p in pose_net._hot_parameters():
        assert float(p._dn_grad_view.abs().max()) > 0.0           # the warp's pose gradient reaches every PoseExpNet layer
    opt.step()
    moved = (opt.arena.flat_p - before).abs()
    assert torch.isfinite(opt.arena.flat_p).all() and 5e-5 < float(moved.max()) <= 1.001e-4


# ================================================================================================ config 5
def _dorn_run(b, x, gt, mask, net, K=80):
    net._dropout_mask = mask.to(DEV)
    tgt = U.get_labels_sid(gt.to(DEV), ordinal_c=K, dataset="kitti")
    dec, ordc = net(x.to(DEV))
    loss = LF.DORN_loss(gt.to(DEV), ordc, tgt, "kitti")
    return tgt, dec, ordc, loss


def test_config5_dorn_ordinal_c_80_vs_reference_golden_and_oracle(golden):
    g = golden("dorn80_cfg")
    # ---- b = 2: the reference's own numbers
    x, gt, mask = dorn80_inputs()
    net = models.Disp_vgg_BN_DORN(datasets="kitti", ordinal_c=80, with_classifier=False)
    sd0 = _fresh(net, "vggdorn80")
    net.to(DEV).train()
    tgt, dec, ordc, loss = _dorn_run(2, x, gt, mask, net)
    loss.backward()
    torch.cuda.synchronize()
    assert dec.dtype == torch.int64 and tuple(dec.shape) == (2, 1, 128, 416) and tuple(ordc.shape) == (2, 80, 128, 416)
    assert abs(int(tgt.long().sum()) - int(g["labels_sum"])) <= 2                          # SID labels (see _labels_match)
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=2e-4)
    _summary_close("ord(golden)", ordc, g, "ord_")
    got = dec.reshape(-1)[::997].cpu().numpy()
    assert (got != g["decode_samples"]).mean() < 5e-3                                      # only P within rounding of 0.5 may flip
    assert abs(int(dec.sum()) - int(g["decode_sum"])) <= 2e-4 * int(g["decode_sum"])
    grad_close("conv_ord.weight grad (golden)", net.conv_ord.weight.grad, g["full:grad:conv_ord.weight"])
    grad_close("conv_ord.bias grad (golden)", net.conv_ord.bias.grad, g["full:grad:conv_ord.bias"])
    # ---- b = 4: the oracle on the same inputs, every parameter
    b = 4
    x, gt, mask = dorn80_inputs(b)
    net = models.Disp_vgg_BN_DORN(datasets="kitti", ordinal_c=80, with_classifier=False)
    sd0 = _fresh(net, "vggdorn80")
    net.to(DEV).train()
    tgt, dec, ordc, loss = _dorn_run(b, x, gt, mask, net)
    loss.backward()
    osd = _oracle_params(sd0)
    odec, oord = ON.disp_vgg_bn_dorn(osd, x, training=True, dropout_mask=mask.view(b, 16, 1, 1))
    otgt = OI.get_labels_sid(gt, ordinal_c=80, dataset="kitti")
    _labels_match(tgt.cpu().numpy(), otgt.numpy(), gt.numpy(), 80)
    oloss = OL.DORN_loss(gt, oord, otgt, "kitti")
    oloss.backward()
    np.testing.assert_allclose(loss.item(), oloss.item(), rtol=2e-4)
    close("ord", ordc, oord, rtol=1e-3, atol_rel=1e-4)
    assert (dec.cpu() != odec).float().mean() < 2e-3
    for name, p in net.named_parameters():
        if _is_pre_bn_conv_bias(name):
            continue
        og = osd[name].grad
        if og is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        grad_close("grad:" + name, p.grad, og)


def test_config5_dorn_full_batch_32x128x416_step():
    B, H, W = 32, 128, 416
    g = torch.Generator().manual_seed(5)
    img = ((torch.rand(B, 3, H, W, generator=g) - 0.5) / 0.5).to(DEV)
    gt = ((torch.rand(B, H, W, generator=g) * 79.0 + 1.0) * (torch.rand(B, H, W, generator=g) < 0.05).float()).to(DEV)
    torch.manual_seed(0)
    net = models.Disp_vgg_BN_DORN(datasets="kitti", ordinal_c=80, with_classifier=False)
    net.init_weights(use_pretrained_weights=False)
    net.to(DEV)
    net.eval()
    with torch.no_grad():
        dec, ordc = net(img)
        dec2, ordc2 = net(img)
        decp, ordp = net(img[8:12].contiguous())
    assert torch.equal(ordc, ordc2) and torch.equal(dec, dec2)                              # deterministic
    assert torch.allclose(ordc[8:12], ordp, rtol=1e-5, atol=1e-6)                          # eval: samples independent
    assert float(ordc.min()) >= 0.0 and float(ordc.max()) <= 1.0
    assert torch.equal(dec, (ordc > 0.5).sum(1, keepdim=True))                             # decode = count of P > 0.5, exact
    depth = U.get_depth_sid(dec, ordinal_c=80, dataset="kitti")
    assert float(depth.min()) >= 0.0 and float(depth.max()) < 81.0
    net.train()
    opt = FusedAdam(net._hot_parameters(), lr=1e-4, betas=(0.9, 0.999), production_order=net._grad_production_order())
    before = opt.arena.flat_p.clone()
    tgt = U.get_labels_sid(gt, ordinal_c=80, dataset="kitti")
    assert int(tgt.min()) >= 0 and int(tgt.max()) <= 80
    _dec, ordc = net(img)
    loss = LF.DORN_loss(gt, ordc, tgt, "kitti")
    opt.zero_grad()
    loss.backward()
    assert np.isfinite(loss.item()) and torch.isfinite(opt.arena.flat_g).all()
    assert float(net.conv_ord.weight._dn_grad_view.abs().max()) > 0
    assert sum(float(p._dn_grad_view.abs().max()) > 0.0 for p in opt.arena.params) >= 0.75 * len(opt.arena.params)
    opt.step()
    moved = (opt.arena.flat_p - before).abs()
    assert torch.isfinite(opt.arena.flat_p).all() and 5e-5 < float(moved.max()) <= 1.001e-4


def test_config5_mixed_precision_mode_vs_fp32():
    """BASELINE configs[4] is the reference's mixed-precision run.  engine.set_compute("bf16") = bf16 multiplies with fp32
    accumulation in the Winograd forward / input-gradient kernels (23 of the 27 convolutions of Disp_vgg_BN_DORN, 91 % of the
    multiply-accumulates); tensors, BatchNorm statistics, weight gradients, the ordinal head, the loss and Adam stay fp32.
    Checked against the fp32 mode of the same network on the same inputs (itself pinned to the reference's golden above).
    STATED TOLERANCES of the mixed-precision mode (13 stacked bf16-product layers, batch statistics renormalise each; measured values
    in brackets): ordinal probabilities max |dP| <= 0.02 (0.0053), mean |dP| <= 1e-3 (2.3e-4); decoded labels identical wherever no
    fp32 probability lies within 0.02 of the 0.5 threshold; loss relative 1e-3 (1e-7); whole gradient vector relative L2 <= 1e-2
    over the decoder + head (1.6e-3) and <= 0.15 over the encoder (0.09: ReLU / max-pool masks of a freshly initialised network flip
    under a 4e-3 perturbation of their inputs, the same mechanism tests/test_gpu_models.py documents between two fp32 kernels)."""
    b = 4
    x, gt, mask = dorn80_inputs(b)
    res = {}
    prev = engine.compute_mode()
    try:
        for mode in ("f32", "bf16"):
            engine.set_compute(mode)
            net = models.Disp_vgg_BN_DORN(datasets="kitti", ordinal_c=80, with_classifier=False)
            _fresh(net, "vggdorn80")
            net.to(DEV).train()
            engine.PROFILE = []
            tgt, dec, ordc, loss = _dorn_run(b, x, gt, mask, net)
            loss.backward()
            torch.cuda.synchronize()
            prof, engine.PROFILE = engine.PROFILE, None
            names = [r[0] for r in prof]
            grads = {n: p.grad.detach().double().cpu() for n, p in net.named_parameters() if p.grad is not None and not _is_pre_bn_conv_bias(n)}
            res[mode] = (dec.cpu(), ordc.detach().cpu(), float(loss.item()), grads, names)
    finally:
        engine.PROFILE = None
        engine.set_compute(prev)
    is_bf = lambda n: "wino_conv_kernel" in n and n.endswith(", 1>")
    nbf = sum(is_bf(n) for n in res["bf16"][4])
    assert nbf >= 30 and not any(is_bf(n) for n in res["f32"][4]), nbf      # the bf16 kernels are what ran
    d0, o0, l0, g0, _ = res["f32"]
    d1, o1, l1, g1, _ = res["bf16"]
    dP = (o1 - o0).abs()
    flips = (d1 != d0).float().mean().item()
    worst = int((d1 - d0).abs().max())
    print("mixed precision vs fp32: max|dP| %.4f mean|dP| %.2e, decode differs on %.3f %% (worst %d bins), loss %.6f vs %.6f" % (
        float(dP.max()), float(dP.mean()), 100 * flips, worst, l1, l0))
    rels = {}
    for block in ("features.", ""):
        names = [n for n in g0 if n.startswith("features.") == (block == "features.")]
        num = sum(float((g1[n] - g0[n]).pow(2).sum()) for n in names)
        den = sum(float(g0[n].pow(2).sum()) for n in names)
        rels[block or "decoder"] = (num / den) ** 0.5
        print("  gradient block %-10s relative L2 %.3e over %d tensors" % (block or "decoder", rels[block or "decoder"], len(names)))
    assert float(dP.max()) <= 0.02 and float(dP.mean()) <= 1e-3
    # decoded label = number of P > 0.5: may only differ where some fp32 probability lies within the stated dP bound of 0.5 (at
    # initialisation that is most of the image: every P starts near 0.5; measured 6.5 % of the pixels differ, by at most 4 bins)
    safe = ((o0 - 0.5).abs() > 0.02).all(1, keepdim=True)
    assert torch.equal(d1[safe], d0[safe])
    assert abs(l1 - l0) <= 1e-3 * abs(l0)
    assert float(dP.max()) > 1e-6                                                          # (bf16 really was used)
    assert rels["features."] <= 0.15 and rels["decoder"] <= 1e-2, rels


# ================================================================================================ config 4
N4, H4, W4 = 16, 480, 640

# name, pieces [(channels, up)], cout, k, stride, pad, reflect, transposed, out_pad, input h, w (of the un-upsampled operands' conv input)
C4_LAYERS = [
    ("stem_7x7s2",          [(3, False)],                   64, 7, 2, 3, False, False, 0, 480, 640),
    ("layer1_1x1_64_256",   [(64, False)],                 256, 1, 1, 0, False, False, 0, 120, 160),
    ("layer1_3x3",          [(64, False)],                  64, 3, 1, 1, False, False, 0, 120, 160),
    ("layer2_3x3s2",        [(128, False)],                128, 3, 2, 1, False, False, 0, 120, 160),
    ("layer2_ds_1x1s2",     [(256, False)],                512, 1, 2, 0, False, False, 0, 120, 160),
    ("layer3_3x3",          [(256, False)],                256, 3, 1, 1, False, False, 0, 30, 40),
    ("layer4_1x1_1024_512", [(1024, False)],               512, 1, 1, 0, False, False, 0, 30, 40),
    ("layer4_3x3s2",        [(512, False)],                512, 3, 2, 1, False, False, 0, 30, 40),
    ("layer4_3x3_odd_map",  [(512, False)],                512, 3, 1, 1, False, False, 0, 15, 20),
    ("layer4_1x1_512_2048", [(512, False)],               2048, 1, 1, 0, False, False, 0, 15, 20),
    ("upconv5_T3x3s2",      [(2048, False)],               256, 3, 2, 1, False, True, 1, 15, 20),
    ("iconv5_cat",          [(256, False), (1024, False)], 256, 3, 1, 1, False, False, 0, 30, 40),
    ("iconv3_cat_disp",     [(64, False), (256, False), (1, True)], 64, 3, 1, 1, False, False, 0, 120, 160),
    ("dec_reflect_2048",    [(2048, False)],               256, 3, 1, 1, True, False, 0, 15, 20),
    ("dec_reflect_up_cat",  [(32, True), (64, False)],      32, 3, 1, 1, True, False, 0, 240, 320),
    ("dec_reflect_up_full", [(16, True)],                   16, 3, 1, 1, True, False, 0, 480, 640),
]


def _ref_points(xs, w, k, stride, pad, reflect, transposed, pts, IH, IW):
    """fp64 value of a few output pixels straight from the definition (nn.Conv2d / nn.ConvTranspose2d over the channel-concatenated,
    optionally nearest-x2 upsampled, optionally reflection-padded input)."""
    wn = w.detach().double().cpu().numpy()
    out = []
    for (n, oy, ox) in pts:
        def pixel(iy, ix):
            if reflect:
                iy = -iy if iy < 0 else (2 * (IH - 1) - iy if iy >= IH else iy)
                ix = -ix if ix < 0 else (2 * (IW - 1) - ix if ix >= IW else ix)
            elif iy < 0 or iy >= IH or ix < 0 or ix >= IW:
                return None
            return np.concatenate([(x[n, iy >> 1, ix >> 1] if up else x[n, iy, ix]).double().cpu().numpy() for x, up in xs])
        acc = 0.0
        for r in range(k):
            for s in range(k):
                if transposed:
                    ty, tx = oy + pad - r, ox + pad - s
                    if ty % stride or tx % stride:
                        continue
                    v = pixel(ty // stride, tx // stride)
                    if v is not None:
                        acc = acc + v @ wn[:, :, r, s]                    # weight [Cin, Cout, k, k]
                else:
                    v = pixel(oy * stride - pad + r, ox * stride - pad + s)
                    if v is not None:
                        acc = acc + wn[:, :, r, s] @ v                    # weight [Cout, Cin, k, k]
        out.append(acc)
    return np.stack(out)


@pytest.mark.parametrize("case", C4_LAYERS, ids=[c[0] for c in C4_LAYERS])
def test_config4_layer_identities_at_16x480x640_shapes(case):
    name, pcs, cout, k, stride, pad, reflect, transposed, out_pad, h, w = case
    cin = sum(c for c, _ in pcs)
    gcpu = torch.Generator(device="cpu").manual_seed(7)
    if transposed:
        mod = nn.ConvTranspose2d(cin, cout, k, stride, pad, out_pad).to(DEV)
    else:
        mod = nn.Conv2d(cin, cout, k, stride, 0 if reflect else pad).to(DEV)
    with torch.no_grad():
        mod.weight.copy_((torch.rand(mod.weight.shape, generator=gcpu) - 0.5) * (2.0 / (k * k * cin) ** 0.5))
        mod.bias.zero_()
    layer = engine.ConvLayer(mod, transposed=transposed, reflect_pad=pad if reflect else 0)
    gen = torch.Generator(device=DEV).manual_seed(11)

    def make_pieces(scale=None):
        acts, pieces = [], []
        for c, up in pcs:
            hh, ww = (h // 2, w // 2) if up else (h, w)
            t = torch.rand(N4, hh, ww, c, device=DEV, generator=gen) - 0.5
            a = engine.Act(t, N4, hh, ww, c)
            a.needs_grad = True
            acts.append(a)
            pieces.append(engine.Piece(a, up=up))
        return acts, pieces

    acts, pieces = make_pieces()
    y, _, _ = engine.conv_forward(layer, pieces, ACT_NONE)
    OH, OW = y.shape[1], y.shape[2]
    assert (OH, OW) == layer.out_size(h, w)
    dy = torch.rand(N4, OH, OW, cout, device=DEV, generator=gen) - 0.5
    engine.conv_dgrad(layer, dy, N4, OH, OW, pieces, (h, w))
    dw = engine.conv_wgrad(layer, pieces, dy, (OH, OW))
    torch.cuda.synchronize()
    a = dot64(dy, y)
    b = sum(dot64(ac.grad, ac.t) for ac in acts)
    c = dot64(dw, mod.weight.detach())
    scale = float(dy.double().norm() * y.double().norm())
    assert abs(a - b) <= 3e-6 * scale, "%s: <dy,conv(x)> %.9g vs <dgrad(dy),x> %.9g" % (name, a, b)
    assert abs(a - c) <= 3e-6 * scale, "%s: <dy,conv(x)> %.9g vs <wgrad,w> %.9g" % (name, a, c)
    # linearity in x
    acts2, pieces2 = make_pieces()
    y2, _, _ = engine.conv_forward(layer, pieces2, ACT_NONE)
    pieces3 = [engine.Piece(engine.Act(0.5 * a1.t - 2.0 * a2.t, a1.N, a1.H, a1.W, a1.C), up=p.up) for a1, a2, p in zip(acts, acts2, pieces)]
    y3, _, _ = engine.conv_forward(layer, pieces3, ACT_NONE)
    lin = float((0.5 * y - 2.0 * y2 - y3).abs().max())
    assert lin <= 3e-5 * float(y.abs().max() + 2 * y2.abs().max()), "%s: forward not linear (%.3g)" % (name, lin)
    # subsampled check against the definition (fp64 on the host), corners and borders included
    r = np.random.RandomState(3)
    pts = [(0, 0, 0), (N4 - 1, OH - 1, OW - 1), (1, 0, OW - 1), (2, OH - 1, 0)] + [(r.randint(N4), r.randint(OH), r.randint(OW)) for _ in range(28)]
    want = _ref_points([(ac.t, p.up) for ac, p in zip(acts, pieces)], mod.weight, k, stride, pad, reflect, transposed, pts, h, w)
    got = np.stack([y[n, oy, ox].double().cpu().numpy() for n, oy, ox in pts])
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=3e-5 * float(np.abs(want).max()), err_msg=name)


def test_config4_pointwise_passes_at_full_size_vs_pytorch_cpu():
    """3x3 / stride 2 / pad 1 max-pool (models/Disp_res_50.py:73) on the 16 x 64 x 240 x 320 stem output, and the bottleneck tail
    relu(bn3(y) + bn_ds(r)) with training-mode statistics on layer1's 16 x 256 x 120 x 160, forward and backward."""
    gen = torch.Generator(device=DEV).manual_seed(2)
    # ---- max-pool
    x = torch.rand(N4, 240, 320, 64, device=DEV, generator=gen) - 0.5
    xa = engine.Act(x, N4, 240, 320, 64)
    tape = engine.Tape(True)
    out = engine.block_maxpool3s2(tape, xa)
    go = torch.rand(out.t.shape, device=DEV, generator=gen) - 0.5
    out.grad = go.clone()
    tape.run_backward()
    xc = x.permute(0, 3, 1, 2).cpu().requires_grad_()
    oc = F.max_pool2d(xc, 3, 2, 1)
    oc.backward(go.permute(0, 3, 1, 2).cpu())
    assert torch.equal(out.t.permute(0, 3, 1, 2).cpu(), oc)
    assert torch.allclose(xa.grad.permute(0, 3, 1, 2).cpu(), xc.grad, rtol=0, atol=1e-6)     # random floats: no ties, same argmax
    del x, xa, out, go, xc, oc
    # ---- bottleneck tail with a downsample branch (both sides carry a pending ReLU-less BatchNorm)
    rows, C = N4 * 120 * 160, 256
    y_t = torch.randn(N4, 120, 160, C, device=DEV, generator=gen)
    r_t = torch.randn(N4, 120, 160, C, device=DEV, generator=gen) * 0.7 + 0.1
    gcpu = torch.Generator().manual_seed(4)
    gam = [(torch.rand(C, generator=gcpu) * 0.4 + 0.8) for _ in range(2)]
    bet = [(torch.rand(C, generator=gcpu) * 0.2 - 0.1) for _ in range(2)]

    def pending(t, gamma, beta):
        a = engine.Act(t, N4, 120, 160, C)
        mean = t.double().mean((0, 1, 2))
        var = t.double().var((0, 1, 2), unbiased=False)
        a.mean, a.invstd = mean.float(), (1.0 / torch.sqrt(var + 1e-5)).float()
        a.scale = (gamma.to(DEV) * a.invstd).contiguous()
        a.shift = (beta.to(DEV) - a.mean * a.scale).contiguous()
        a.no_relu = True
        return a

    ya, ra = pending(y_t, gam[0], bet[0]), pending(r_t, gam[1], bet[1])
    tape = engine.Tape(True)
    out = engine.block_bn_add_relu(tape, ya, ra)
    go = torch.rand(out.t.shape, device=DEV, generator=gen) - 0.5
    out.grad = go.clone()
    tape.run_backward()
    torch.cuda.synchronize()
    # PyTorch-CPU: d out / d(bn output) only (the BatchNorm backward itself is the next kernel, dn_bn_bwd_apply, checked below)
    yc, rc = y_t.permute(0, 3, 1, 2).cpu(), r_t.permute(0, 3, 1, 2).cpu()
    zy = F.batch_norm(yc, None, None, gam[0], bet[0], True, 0.0, 1e-5).requires_grad_()
    zr = F.batch_norm(rc, None, None, gam[1], bet[1], True, 0.0, 1e-5).requires_grad_()
    oc = torch.relu(zy + zr)
    oc.backward(go.permute(0, 3, 1, 2).cpu())
    assert torch.allclose(out.t.permute(0, 3, 1, 2).cpu(), oc, rtol=1e-4, atol=2e-4)   # batch statistics here in fp64, there in fp32
    assert ya.grad_is_dz and ra.grad_is_dz
    # the tail's backward hands dL/dz (the ReLU-masked gradient) to both branches; masks may differ where |z| ~ rounding
    for a_, z_ in ((ya, zy), (ra, zr)):
        got = a_.grad.permute(0, 3, 1, 2).cpu()
        bad = (got - z_.grad).abs() > 1e-6
        assert bad.float().mean() < 1e-5
    # column sums produced alongside (dbeta, dgamma-precursor) against the CPU in fp64
    part = ya.partial.double().sum(0)                                                       # [C][4]
    # the sums are over the dz the kernel itself wrote (checked element-wise above): a ReLU mask that flips where |z| is within
    # rounding of zero moves a column sum by one whole |go| <= 0.5, which is not a summation error
    dz = ya.grad.permute(0, 3, 1, 2).cpu().double()
    xhat = ((yc.double() - yc.double().mean((0, 2, 3), keepdim=True)) / torch.sqrt(yc.double().var((0, 2, 3), unbiased=False, keepdim=True) + 1e-5))
    want_db = dz.sum((0, 2, 3))
    want_dg = (dz * xhat).sum((0, 2, 3))
    sc = float(want_db.abs().max())
    assert torch.allclose(part[:, 0].cpu(), want_db, rtol=1e-3, atol=1e-4 * sc)
    assert torch.allclose(part[:, 1].cpu(), want_dg, rtol=1e-3, atol=1e-4 * float(want_dg.abs().max()))


@pytest.mark.parametrize("which", ["Disp_res_50", "monodepth2_res50_depthdecoder"])
def test_config4_full_step_16x480x640(which):
    g = torch.Generator().manual_seed(0)
    img = ((torch.rand(N4, 3, H4, W4, generator=g) - 0.5) / 0.5).to(DEV)
    gt = (torch.rand(N4, H4, W4, generator=g) * 9.5 + 0.5).to(DEV)
    torch.manual_seed(0)
    if which == "Disp_res_50":
        net = models.Disp_res_50(datasets="nyu")
        net.init_weights(use_pretrained_weights=False)
        order = net._grad_production_order()
        scale, shapes = 1.0, [(N4, 1, 480, 640), (N4, 1, 240, 320), (N4, 1, 120, 160), (N4, 1, 60, 80)]
        lo, hi = 0.0999, 10.1001
    else:
        enc = networks.ResnetEncoder(50, False)
        net = models.monodepth2(enc, networks.DepthDecoder(enc.num_ch_enc))
        order = None
        img = (img + 1) / 2                                                               # monodepth2 nets take [0,1] images
        shapes = [(N4, 1, 480, 640), (N4, 1, 240, 320), (N4, 1, 120, 160), (N4, 1, 60, 80)]
        lo, hi = 0.0099, 10.0001
    net.to(DEV)
    net.eval()
    with torch.no_grad():
        full = net(img)
        again = net(img)
        part = net(img[3:5].contiguous())
    assert full.shape == (N4, 1, H4, W4) and torch.equal(full, again)
    assert torch.allclose(full[3:5], part, rtol=1e-4, atol=1e-5)                           # eval: samples independent of the batch
    assert float(full.min()) > lo and float(full.max()) < hi
    net.train()
    opt = FusedAdam(net._hot_parameters(), lr=1e-4, betas=(0.9, 0.999), production_order=order)
    before = opt.arena.flat_p.clone()
    disps = net(img)
    assert [tuple(d.shape) for d in disps] == shapes
    depth = [reciprocal(d) for d in disps]
    loss = LF.l1_loss(gt, depth, "nyu") + 0.1 * LF.smooth_loss(depth)
    opt.zero_grad()
    loss.backward()
    assert np.isfinite(loss.item()) and torch.isfinite(opt.arena.flat_g).all()
    dead = [p for p in opt.arena.params if p.dim() > 1 and float(p._dn_grad_view.abs().max()) == 0.0]
    assert not dead, "%d weight tensors without gradient" % len(dead)
    opt.step()
    moved = (opt.arena.flat_p - before).abs()
    assert torch.isfinite(opt.arena.flat_p).all() and 5e-5 < float(moved.max()) <= 1.001e-4
    # training-mode BatchNorm bookkeeping at this size: running statistics moved off their defaults exactly once
    bns = [m for m in net.modules() if isinstance(m, nn.BatchNorm2d)]
    assert bns and all(int(m.num_batches_tracked) == 1 for m in bns)
