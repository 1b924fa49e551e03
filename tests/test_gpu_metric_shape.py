"""Parity AT the benchmarked shape: the network tests elsewhere compare with the oracle at b2 (64x96, 128x416),
but kernel selection depends on the grid -- the 8-wave Winograd kernel vs the 4-wave one, the tail split at 832 blocks, K splits, the
lds3k roles --, so b2 does not exercise the dispatch of the literal metric config.  Here the CPU oracle runs Disp_vgg_BN (configs[1]) and
the K = 80 ordinal net (configs[4]) at 32 x 128 x 416 on the host cores (a few seconds on the GPU box's 128 cores; torch threads are
capped at 32, the count that is fastest there) and the HIP path is compared with it on identical closed-form inputs:

  loss                        rtol 1e-4
  the four disparity maps     every 97th element, rtol 1e-3 / atol 1e-4 * max|ref|  (+ sum rtol 1e-4)
  parameter gradients         grad_close (tests/test_gpu_models.py: relative L2 <= 2e-2, <= 1 % outliers, scale coefficient within 3e-3)
                              on tensors spanning encoder stage 1, stage 5 and the decoder
in the default arithmetic (f32x3) AND with the fp32 matrix instruction (f32).  Also here: the reference's nn.DataParallel wrapper
(train.py:316-317,378) around a drop-in module on one device.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import supervised_dispnet_amd.loss_functions as LF  # noqa: E402
import supervised_dispnet_amd.models as models  # noqa: E402
import supervised_dispnet_amd.utils as U  # noqa: E402
from oracle import detgen, image_ops as OI, losses as OL, nets as ON  # noqa: E402  (the checker)
from supervised_dispnet_amd import engine  # noqa: E402
from supervised_dispnet_amd.functional import reciprocal  # noqa: E402
from test_gpu_models import _is_pre_bn_conv_bias, _oracle_params, close, grad_close  # noqa: E402

DEV = torch.device("cuda:0")
B, H, W = 32, 128, 416
_ORACLE = {}


class _host_threads(object):
    """The oracle on at most 32 torch threads: every logical CPU of the 2-socket GPU host in one oneDNN pool is ~200x slower."""

    def __enter__(self):
        self.prev = torch.get_num_threads()
        torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))

    def __exit__(self, *exc):
        torch.set_num_threads(self.prev)
        return False


def _vgg_oracle():
    if "vgg" not in _ORACLE:
        net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
        detgen.fill_state_dict(net.state_dict(), "vggbn")
        sd0 = {k: v.clone() for k, v in net.state_dict().items()}
        x = detgen.image_batch(B, H, W, "metric32:x")
        gt = detgen.sparse_depth(B, H, W, "metric32:gt", density=0.05)
        osd = _oracle_params(sd0)
        with _host_threads():
            odisps = ON.disp_vgg_bn(osd, x, training=True)
            oloss = OL.l1_loss(gt, [1 / d for d in odisps], "kitti")
            oloss.backward()
        # (F.batch_norm in training mode updated the running statistics of the oracle's dict in place: one training forward, like the net's)
        _ORACLE["vgg"] = (sd0, x, gt, [d.detach() for d in odisps], float(oloss.item()),
                          {k: v.grad.clone() for k, v in osd.items() if getattr(v, "grad", None) is not None},
                          {k: v.detach().clone() for k, v in osd.items() if "running" in k})
    return _ORACLE["vgg"]


# encoder stage 1 (conv1_1 on the stem kernel, conv1_2 = the 64-channel full-resolution Winograd layer + its BatchNorm), stage 3,
# stage 5 (the deep 512-channel layers: small grids, K splits), decoder (transposed convolutions, virtual concats, lds3 / lds3k, heads)
VGG_GRAD_KEYS = ("features.features.0.weight", "features.features.3.weight", "features.features.4.weight", "features.features.4.bias",
                 "features.features.17.weight", "features.features.34.weight", "features.features.40.weight", "features.features.41.weight",
                 "upconv4.0.weight", "iconv4.0.weight", "iconv2.0.weight", "upconv1.0.weight", "iconv1.0.weight", "iconv0.0.weight", "iconv0.0.bias",
                 "disp0.0.weight", "disp0.0.bias", "disp2.0.weight")


@pytest.mark.parametrize("mode", ["f32x3", "f32"])
def test_disp_vgg_bn_at_32x128x416_vs_oracle(mode):
    sd0, x, gt, odisps, oloss, ograds, obn = _vgg_oracle()
    prev = engine.compute_mode()
    engine.set_compute(mode)
    try:
        net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
        net.load_state_dict(sd0)
        net.to(DEV).train()
        disps = net(x.to(DEV))
        loss = LF.l1_loss(gt.to(DEV), [reciprocal(d) for d in disps], "kitti")
        loss.backward()
        torch.cuda.synchronize()
    finally:
        engine.set_compute(prev)
    np.testing.assert_allclose(loss.item(), oloss, rtol=1e-4)
    for i, (d, od) in enumerate(zip(disps, odisps)):
        assert tuple(d.shape) == tuple(od.shape)
        got, want = d.detach().reshape(-1)[::97].cpu(), od.reshape(-1)[::97]
        close("%s disp%d[::97]" % (mode, i), got, want, rtol=1e-3, atol_rel=1e-4)
        np.testing.assert_allclose(float(d.double().sum()), float(od.double().sum()), rtol=1e-4)
    named = dict(net.named_parameters())
    for key in VGG_GRAD_KEYS:
        grad_close("%s grad:%s" % (mode, key), named[key].grad, ograds[key])
    for name, p in named.items():
        if _is_pre_bn_conv_bias(name):
            assert float(p.grad.abs().max()) == 0.0
        elif name in ograds:
            assert p.grad is not None and torch.isfinite(p.grad).all(), name
    # BatchNorm running statistics of the first and the last encoder layer after this ONE training forward
    sd1 = net.state_dict()
    for key in ("features.features.1.running_mean", "features.features.1.running_var", "features.features.41.running_mean",
                "features.features.41.running_var"):
        close(mode + " " + key, sd1[key], obn[key], rtol=1e-3, atol_rel=1e-4)


def test_config5_k80_head_at_32x128x416_vs_oracle():
    """configs[4] at the literal batch: SID labels, the fused ordinal head (ord_c1, decode), DORN_loss and the gradients of the head and
    of encoder / decoder tensors against the oracle (tolerances of test_config5_dorn_ordinal_c_80_vs_reference_golden_and_oracle)."""
    from cases import dorn80_inputs
    from test_gpu_fullsize_configs import _labels_match
    K = 80
    x, gt, mask = dorn80_inputs(B)
    net = models.Disp_vgg_BN_DORN(datasets="kitti", ordinal_c=K, with_classifier=False)
    detgen.fill_state_dict(net.state_dict(), "vggdorn80")
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    net.to(DEV).train()
    net._dropout_mask = mask.to(DEV)
    tgt = U.get_labels_sid(gt.to(DEV), ordinal_c=K, dataset="kitti")
    dec, ordc = net(x.to(DEV))
    loss = LF.DORN_loss(gt.to(DEV), ordc, tgt, "kitti")
    loss.backward()
    torch.cuda.synchronize()
    osd = _oracle_params(sd0)
    with _host_threads():
        odec, oord = ON.disp_vgg_bn_dorn(osd, x, training=True, dropout_mask=mask.view(B, 16, 1, 1))
        otgt = OI.get_labels_sid(gt, ordinal_c=K, dataset="kitti")
        oloss = OL.DORN_loss(gt, oord, otgt, "kitti")
        oloss.backward()
    _labels_match(tgt.cpu().numpy(), otgt.numpy(), gt.numpy(), K)
    np.testing.assert_allclose(loss.item(), oloss.item(), rtol=2e-4)
    close("ord[::97]", ordc.detach().reshape(-1)[::97].cpu(), oord.detach().reshape(-1)[::97], rtol=1e-3, atol_rel=1e-4)
    assert (dec.cpu() != odec).float().mean() < 2e-3               # only probabilities within rounding of 0.5 may flip the count
    named = dict(net.named_parameters())
    for key in ("conv_ord.weight", "conv_ord.bias", "iconv0.0.weight", "upconv0.0.weight", "iconv2.0.weight", "upconv4.0.weight",
                "features.features.3.weight", "features.features.40.weight"):
        grad_close("grad:" + key, named[key].grad, osd[key].grad)


def test_dataparallel_wrapper_on_one_device_equals_the_bare_module():
    """Reference train.py:316-317 (`disp_net = torch.nn.DataParallel(disp_net)`) and :378 (`disp_net.module.state_dict()`): on a
    one-GPU box the wrapper calls the module itself.  Forward outputs, loss, every gradient and the state_dict under `.module` equal
    the bare module's, bit for bit; eval mode returns disp0 alone through the wrapper too.  (Beyond one device the models refuse
    replication with a message: tests/test_cli_and_host.py::test_models_refuse_multi_device_dataparallel_replication.)"""
    x = detgen.image_batch(2, 64, 96, "dp:x").to(DEV)
    gt = detgen.sparse_depth(2, 64, 96, "dp:gt", density=0.3).to(DEV)
    res = {}
    for wrapped in (False, True):
        net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
        detgen.fill_state_dict(net.state_dict(), "vggbn")
        net.to(DEV)
        call = torch.nn.DataParallel(net, device_ids=[0]) if wrapped else net
        call.train()
        disps = call(x)
        assert isinstance(disps, tuple) and len(disps) == 4
        loss = LF.l1_loss(gt, [reciprocal(d) for d in disps], "kitti") + 0.1 * LF.smooth_loss([reciprocal(d) for d in disps])
        loss.backward()
        inner = call.module if wrapped else call
        sd = inner.state_dict()
        call.eval()
        with torch.no_grad():
            e = call(x)
        assert e.shape == (2, 1, 64, 96)
        res[wrapped] = (loss.item(), [d.detach().clone() for d in disps], {n: p.grad.clone() for n, p in inner.named_parameters() if p.grad is not None},
                        {k: v.clone() for k, v in sd.items()}, e.clone())
    assert res[True][0] == res[False][0]
    assert all(torch.equal(a, b) for a, b in zip(res[True][1], res[False][1]))
    assert res[True][2].keys() == res[False][2].keys() and len(res[True][2]) > 50
    assert all(torch.equal(res[True][2][k], res[False][2][k]) for k in res[True][2])
    assert res[True][3].keys() == res[False][3].keys() and "features.features.0.weight" in res[True][3]
    assert all(torch.equal(res[True][3][k], res[False][3][k]) for k in res[True][3])
    assert torch.equal(res[True][4], res[False][4])
    # the default device_ids of a one-GPU box is that one device as well
    if torch.cuda.device_count() == 1:
        net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
        detgen.fill_state_dict(net.state_dict(), "vggbn")
        dp = torch.nn.DataParallel(net.to(DEV))
        dp.train()
        assert all(torch.equal(a, b) for a, b in zip(dp(x), res[False][1]))


def test_config3_at_32x128x416_vs_oracle():
    """configs[2] at the literal batch: Disp_vgg_BN + PoseExpNet + photometric warp loss + smoothness, 32 x 128 x 416, against the oracle on
    the host CPU: both loss terms rtol 2e-4, the pose vectors rtol 1e-3, disparity samples, and gradients of both nets through grad_close
    (the b2 test pins every tensor; here tensors spanning the warp -> pose path and the encoder / decoder at the benchmarked dispatch)."""
    from cases import config3_inputs
    from oracle import nets_res
    tgt, refs, k, kinv = config3_inputs(B)
    disp_net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
    detgen.fill_state_dict(disp_net.state_dict(), "vggbn")
    dsd0 = {kk: v.clone() for kk, v in disp_net.state_dict().items()}
    pose_net = models.PoseExpNet(nb_ref_imgs=2, output_exp=False)
    detgen.fill_state_dict(pose_net.state_dict(), "posenet")
    psd0 = {kk: v.clone() for kk, v in pose_net.state_dict().items()}
    disp_net.to(DEV).train()
    pose_net.to(DEV).train()
    tg, rf = tgt.to(DEV), [r.to(DEV) for r in refs]
    mask, pose = pose_net(tg, rf)
    disps = disp_net(tg)
    depth = [reciprocal(d) for d in disps]
    l1 = LF.photometric_reconstruction_loss(tg, rf, k.to(DEV), kinv.to(DEV), depth, mask, pose, "euler", "zeros")
    l3 = LF.smooth_loss(depth)
    (l1 + 0.1 * l3).backward()
    torch.cuda.synchronize()
    dsd, psd = _oracle_params(dsd0), _oracle_params(psd0)
    with _host_threads():
        omask, opose = nets_res.pose_exp_net(psd, tgt, refs, False, training=True)
        odisps = ON.disp_vgg_bn(dsd, tgt, training=True)
        odepth = [1 / d for d in odisps]
        ol1 = OL.photometric_reconstruction_loss(tgt, refs, k, kinv, odepth, omask, opose, "euler", "zeros")
        ol3 = OL.smooth_loss(odepth)
        (ol1 + 0.1 * ol3).backward()
    np.testing.assert_allclose(l1.item(), ol1.item(), rtol=2e-4)
    np.testing.assert_allclose(l3.item(), ol3.item(), rtol=2e-4)
    close("pose", pose, opose, rtol=1e-3, atol_rel=1e-4)
    for i, (d, od) in enumerate(zip(disps, odisps)):
        close("disp%d[::97]" % i, d.detach().reshape(-1)[::97].cpu(), od.detach().reshape(-1)[::97], rtol=1e-3, atol_rel=1e-4)
    dn = dict(disp_net.named_parameters())
    for key in ("features.features.0.weight", "features.features.3.weight", "features.features.40.weight", "upconv4.0.weight", "iconv2.0.weight",
                "iconv0.0.weight", "disp0.0.weight", "disp3.0.weight"):
        grad_close("disp grad:" + key, dn[key].grad, dsd[key].grad)
    pn = dict(pose_net.named_parameters())
    for key in ("conv1.0.weight", "conv4.0.weight", "conv7.0.weight", "pose_pred.weight", "pose_pred.bias"):
        grad_close("pose grad:" + key, pn[key].grad, psd[key].grad)


def test_config4_at_16x480x640_vs_oracle():
    """configs[3] at the literal batch: Disp_res_50 (7x7/2 stem, bottlenecks, residual tails, 3x3/2 transposed-convolution decoder) at
    16 x 480 x 640 with the NYU loss against the oracle on the host CPU: loss rtol 2e-4, four disparity maps (every 997th element) rtol
    2e-3, gradient tensors from the stem to the heads through grad_close, BatchNorm running statistics."""
    from oracle import nets_res
    b, h, w = 16, 480, 640
    net = models.Disp_res_50(datasets="nyu")
    detgen.fill_state_dict(net.state_dict(), "res50")
    sd0 = {kk: v.clone() for kk, v in net.state_dict().items()}
    net.to(DEV).train()
    x = detgen.image_batch(b, h, w, "res50big:x")
    gt = detgen.sparse_depth(b, h, w, "res50big:gt", density=0.6, lo=0.3, hi=11.0)
    disps = net(x.to(DEV))
    depth = [reciprocal(d) for d in disps]
    loss = LF.l1_loss(gt.to(DEV), depth, "nyu") + 0.1 * LF.smooth_loss(depth)
    loss.backward()
    torch.cuda.synchronize()
    osd = _oracle_params(sd0)
    with _host_threads():
        odisps = nets_res.disp_res_50(osd, x, training=True, datasets="nyu")
        odepth = [1 / d for d in odisps]
        oloss = OL.l1_loss(gt, odepth, "nyu") + 0.1 * OL.smooth_loss(odepth)
        oloss.backward()
    np.testing.assert_allclose(loss.item(), oloss.item(), rtol=2e-4)
    for i, (d, od) in enumerate(zip(disps, odisps)):
        assert tuple(d.shape) == tuple(od.shape)
        close("disp%d[::997]" % i, d.detach().reshape(-1)[::997].cpu(), od.detach().reshape(-1)[::997], rtol=2e-3, atol_rel=2e-4)
    named = dict(net.named_parameters())
    for key in ("conv1.weight", "layer1.0.conv1.weight", "layer1.0.downsample.0.weight", "layer2.0.conv2.weight", "layer3.5.conv3.weight",
                "layer4.2.conv2.weight", "layer4.2.bn3.weight", "upconv5.0.weight", "iconv5.0.weight", "iconv2.0.weight", "iconv1.0.weight",
                "predict_disp1.0.weight", "predict_disp4.0.weight"):
        grad_close("grad:" + key, named[key].grad, osd[key].grad)
    sd1 = net.state_dict()
    for key in ("bn1.running_mean", "bn1.running_var", "layer4.2.bn3.running_mean", "layer1.0.downsample.1.running_var"):
        close(key, sd1[key], osd[key], rtol=1e-3, atol_rel=1e-4)
