"""Oracle (torch-CPU) restatement of the rest of the reference's DispNet zoo.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Functions of a state_dict keyed exactly like the reference modules'
state_dict(); BatchNorm running statistics are updated in place in training mode.  Pinned to the imported reference by
tests/golden/zoo.npz (tests/golden/make_goldens.py::gold_zoo) through tests/test_oracle_golden.py::test_model_zoo.

Reference: models/Disp_res_18.py:50-210 + :249-286 (BasicBlock), models/Disp_res.py:59-208, models/Disp_res_101.py:43-196,
models/Disp_vgg.py:71-207, models/Disp_vgg_feature.py:72-192, models/FCRN.py:52-260, models/ASPP.py:53-196, models/res_aspp.py (same text).
"""
import torch
import torch.nn.functional as F

from .nets import _bn, _crop_like, _head, _nearest2, alpha_beta
from .nets_res import _conv3_lrelu, _up3_lrelu, res_layers


def _bil2(x):
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)


def disp_res_18(sd, x, training=True, datasets="kitti", only_train_dec=False):
    """models/Disp_res_18.py:138-210: Disp_res_50's forward on BasicBlocks (2, 2, 2, 2), expansion 1; bn1 output discarded (:141-145)."""
    alpha, beta = alpha_beta(datasets)
    conv1 = F.conv2d(x, sd["conv1.weight"], stride=2, padding=3)
    _bn(sd, "bn1", conv1, training)
    relu1 = F.relu(conv1)
    pool1 = F.max_pool2d(relu1, 3, 2, 1)
    conv2, conv3, conv4, conv5 = res_layers(sd, pool1, "", "basic", (2, 2, 2, 2), training)
    if only_train_dec:
        relu1, conv2, conv3, conv4, conv5 = [t.detach() for t in (relu1, conv2, conv3, conv4, conv5)]
    i5 = _conv3_lrelu(sd, "iconv5", torch.cat((_up3_lrelu(sd, "upconv5", conv5), conv4), 1))
    i4 = _conv3_lrelu(sd, "iconv4", torch.cat((_up3_lrelu(sd, "upconv4", i5), conv3), 1))
    d4 = _head(sd, "predict_disp4", i4, alpha, beta)
    i3 = _conv3_lrelu(sd, "iconv3", torch.cat((_up3_lrelu(sd, "upconv3", i4), conv2, _nearest2(d4)), 1))
    d3 = _head(sd, "predict_disp3", i3, alpha, beta)
    i2 = _conv3_lrelu(sd, "iconv2", torch.cat((_up3_lrelu(sd, "upconv2", i3), relu1, _nearest2(d3)), 1))
    d2 = _head(sd, "predict_disp2", i2, alpha, beta)
    i1 = _conv3_lrelu(sd, "iconv1", torch.cat((_up3_lrelu(sd, "upconv1", i2), _nearest2(d2)), 1))
    d1 = _head(sd, "predict_disp1", i1, alpha, beta)
    return (d1, d2, d3, d4) if training else d1


def disp_res6(sd, x, training=True, datasets="kitti", only_train_dec=False, layer3_blocks=6, leaky=True):
    """models/Disp_res.py:137-208 (layer3_blocks=6, leaky=True) and models/Disp_res_101.py:129-196 (23, False): six decoder
    levels, crop_like after every transposed convolution, bilinear x2 of the disparities.  skip2 = pool1 is at H/4 while upconv3's
    result is at H/2: the crop keeps its top-left quarter (disp3 comes out at H/4)."""
    alpha, beta = alpha_beta(datasets)
    act = (lambda t: F.leaky_relu(t, 0.1)) if leaky else F.relu
    up = lambda name, t: act(F.conv_transpose2d(t, sd[name + ".0.weight"], sd[name + ".0.bias"], stride=2, padding=1, output_padding=1))
    ic = lambda name, t: act(F.conv2d(t, sd[name + ".0.weight"], sd[name + ".0.bias"], padding=1))
    conv1 = F.conv2d(x, sd["conv1.weight"], stride=2, padding=3)
    _bn(sd, "bn1", conv1, training)
    relu1 = F.relu(conv1)
    pool1 = F.max_pool2d(relu1, 3, 2, 1)
    conv2, conv3, conv4, conv5 = res_layers(sd, pool1, "", "bottleneck", (3, 4, layer3_blocks, 3), training)
    if only_train_dec:
        relu1, pool1, conv2, conv3, conv4, conv5 = [t.detach() for t in (relu1, pool1, conv2, conv3, conv4, conv5)]
    skip1, skip2, skip3, skip4, skip5 = relu1, pool1, conv2, conv3, conv4
    i6 = ic("iconv6", torch.cat((_crop_like(up("upconv6", conv5), skip5), skip5), 1))
    i5 = ic("iconv5", torch.cat((_crop_like(up("upconv5", i6), skip4), skip4), 1))
    i4 = ic("iconv4", torch.cat((_crop_like(up("upconv4", i5), skip3), skip3), 1))
    d4 = _head(sd, "predict_disp4", i4, alpha, beta)
    i3 = ic("iconv3", torch.cat((_crop_like(up("upconv3", i4), skip2), skip2, _crop_like(_bil2(d4), skip2)), 1))
    d3 = _head(sd, "predict_disp3", i3, alpha, beta)
    i2 = ic("iconv2", torch.cat((_crop_like(up("upconv2", i3), skip1), skip1, _crop_like(_bil2(d3), skip1)), 1))
    d2 = _head(sd, "predict_disp2", i2, alpha, beta)
    i1 = ic("iconv1", torch.cat((_crop_like(up("upconv1", i2), x), _crop_like(_bil2(d2), x)), 1))
    d1 = _head(sd, "predict_disp1", i1, alpha, beta)
    return (d1, d2, d3, d4) if training else d1


VGG_PLAIN_STAGES = {
    # stage -> conv parameter prefixes: models/Disp_vgg.py:96-100 (own Sequentials) / Disp_vgg_feature.py:138-142 (vgg16 slices)
    "Disp_vgg": [["conv1.0", "conv1.2"], ["conv2.0", "conv2.2"], ["conv3.0", "conv3.2", "conv3.4"], ["conv4.0", "conv4.2", "conv4.4"],
                 ["conv5.0", "conv5.2", "conv5.4"]],
    "Disp_vgg_feature": [["features.features.%d" % i for i in idx] for idx in ((0, 2), (5, 7), (10, 12, 14), (17, 19, 21), (24, 26, 28))],
}


def disp_vgg(sd, x, training=True, alpha=10, beta=0.01, layout="Disp_vgg", detach_encoder=False):
    """models/Disp_vgg.py:156-207 == models/Disp_vgg_feature.py:137-192: conv3x3+ReLU stages each ending in MaxPool2d(2,2), ReLU
    decoder, `upsample_nn_nearest` there IS bilinear (:8-9).  detach_encoder = use_pretrained_weights and only_train_dec (:163)."""
    feats, h = [], x
    for stage in VGG_PLAIN_STAGES[layout]:
        for p in stage:
            h = F.relu(F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1))
        h = F.max_pool2d(h, 2, 2)
        feats.append(h)
    if detach_encoder:
        feats = [f.detach() for f in feats]
    c1, c2, c3, c4, c5 = feats
    up = lambda name, t: F.relu(F.conv_transpose2d(t, sd[name + ".0.weight"], sd[name + ".0.bias"], stride=2, padding=1))
    ic = lambda name, t: F.relu(F.conv2d(t, sd[name + ".0.weight"], sd[name + ".0.bias"], padding=1))
    i4 = ic("iconv4", torch.cat((up("upconv4", c5), c4), 1))
    i3 = ic("iconv3", torch.cat((up("upconv3", i4), c3), 1))
    d3 = _head(sd, "disp3", i3, alpha, beta)
    i2 = ic("iconv2", torch.cat((up("upconv2", i3), c2, _bil2(d3)), 1))
    d2 = _head(sd, "disp2", i2, alpha, beta)
    i1 = ic("iconv1", torch.cat((up("upconv1", i2), c1, _bil2(d2)), 1))
    d1 = _head(sd, "disp1", i1, alpha, beta)
    i0 = ic("iconv0", torch.cat((up("upconv0", i1), _bil2(d1)), 1))
    d0 = _head(sd, "disp0", i0, alpha, beta)
    return (d0, d1, d2, d3) if training else d0


# ------------------------------------------------------------------------------------------------------ FCRN
def _upproject(sd, p, x, training):
    """models/FCRN.py:74-124 (UpProject.forward): per branch four un-padded convolutions over hand-padded inputs -- 3x3 on pad (l1, r1, t1,
    b1); 2x3 on (1, 1, 1, 0); 3x2 on (1, 0, 1, 1); 2x2 on (1, 0, 1, 0): the "author's interleaving padding" the reference keeps --,
    interleaved along the width (conv*_1 | conv*_2 and conv*_3 | conv*_4), then along the height; branch 1: BN, ReLU, conv3x3 pad 1, BN;
    branch 2: BN; sum; ReLU."""
    pads = ((1, 1, 1, 1), (1, 1, 1, 0), (1, 0, 1, 1), (1, 0, 1, 0))

    def branch(br):
        o = [F.conv2d(F.pad(x, pads[k]), sd["%s.conv%d_%d.weight" % (p, br, k + 1)], sd["%s.conv%d_%d.bias" % (p, br, k + 1)]) for k in range(4)]
        b, c, h, w = o[0].shape
        top = torch.stack((o[0], o[1]), dim=-1).reshape(b, c, h, 2 * w)          # columns 2x | 2x+1
        bot = torch.stack((o[2], o[3]), dim=-1).reshape(b, c, h, 2 * w)
        return torch.stack((top, bot), dim=-2).reshape(b, c, 2 * h, 2 * w)       # rows 2y | 2y+1

    out1 = F.relu(_bn(sd, p + ".bn1_1", branch(1), training))
    out1 = _bn(sd, p + ".bn2", F.conv2d(out1, sd[p + ".conv3.weight"], sd[p + ".conv3.bias"], padding=1), training)
    out2 = _bn(sd, p + ".bn1_2", branch(2), training)
    return F.relu(out1 + out2)


def fcrn(sd, x, training=True, datasets="kitti", dropout_mask=None):
    """models/FCRN.py:228-260: conv7x7/2-BN-ReLU, max-pool 3/2/1, ResNet-50 layers, conv1x1 2048->1024 + BN (no ReLU), four up-projections,
    Dropout2d (training; `dropout_mask` [N,64] = keep / (1-p) pattern injected for reproducibility), conv3x3 64->1, alpha*sigmoid+beta,
    bilinear resize to the input size with align_corners=True.  Training mode returns a one-element tuple."""
    alpha, beta = alpha_beta(datasets)
    inp = x.shape[2:]
    h = F.relu(_bn(sd, "bn1", F.conv2d(x, sd["conv1.weight"], stride=2, padding=3), training))
    h = F.max_pool2d(h, 3, 2, 1)
    h = res_layers(sd, h, "", "bottleneck", (3, 4, 6, 3), training)[-1]
    h = _bn(sd, "bn2", F.conv2d(h, sd["conv2.weight"]), training)
    for name in ("up1", "up2", "up3", "up4"):
        h = _upproject(sd, name, h, training)
    if training:
        if dropout_mask is None:
            h = F.dropout2d(h, 0.5, True)
        else:
            h = h * dropout_mask.to(h.dtype).view(h.shape[0], h.shape[1], 1, 1)
    h = alpha * torch.sigmoid(F.conv2d(h, sd["conv3.weight"], sd["conv3.bias"], padding=1)) + beta
    h = F.interpolate(h, size=inp, mode="bilinear", align_corners=True)
    return (h,) if training else h


# ------------------------------------------------------------------------------------------------------ ASPP
def _aspp_bottleneck(sd, p, x, stride, dil, training):
    """models/ASPP.py:84-105: 1x1 (STRIDE here)-BN-ReLU, 3x3 dilated (padding = dilation)-BN-ReLU, 1x1-BN, optional downsample (1x1 stride + BN), add, ReLU."""
    out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], stride=stride), training))
    out = F.relu(_bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], padding=dil, dilation=dil), training))
    out = _bn(sd, p + ".bn3", F.conv2d(out, sd[p + ".conv3.weight"]), training)
    identity = x
    if (p + ".downsample.0.weight") in sd:
        identity = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), training)
    return F.relu(out + identity)


def aspp_depth(sd, x, training=True, counts=(3, 4, 6, 3)):
    """models/ASPP.py:127-196 (deeplab_depth, counts (3, 4, 23, 3)) / models/res_aspp.py (res50_aspp, (3, 4, 6, 3)): conv7x7/2-BN-ReLU,
    max-pool 3/2/1 with ceil_mode, layer1, layer2 (stride 2), layer3 (dilation 2), layer4 (dilation 4), the classifier = sum of four
    dilated (6, 12, 18, 24) 3x3 convolutions 2048 -> 1, 10 * sigmoid + 0.01, bilinear resize to the input size (align_corners=True)."""
    inp = x.shape[2:]
    P = "Scale."
    h = F.relu(_bn(sd, P + "bn1", F.conv2d(x, sd[P + "conv1.weight"], stride=2, padding=3), training))
    h = F.max_pool2d(h, 3, 2, 1, ceil_mode=True)
    for li, (n, stride, dil) in enumerate(zip(counts, (1, 2, 1, 1), (1, 1, 2, 4)), start=1):
        for bi in range(n):
            h = _aspp_bottleneck(sd, "%slayer%d.%d" % (P, li, bi), h, stride if bi == 0 else 1, dil, training)
    out = None
    for i, d in enumerate((6, 12, 18, 24)):
        o = F.conv2d(h, sd["%slayer5.conv2d_list.%d.weight" % (P, i)], sd["%slayer5.conv2d_list.%d.bias" % (P, i)], padding=d, dilation=d)
        out = o if out is None else out + o
    out = 10 * torch.sigmoid(out) + 0.01
    out = F.interpolate(out, size=inp, mode="bilinear", align_corners=True)
    return (out,) if training else out
