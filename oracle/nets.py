"""Oracle (torch-CPU, fp32/fp64) restatement of the reference encoder-decoders.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The nets are *functions of a
state_dict* keyed exactly like the reference modules' state_dict(), so a product model's
`state_dict()` can be fed straight in.  BatchNorm running statistics in the dict are
updated in place in training mode, like nn.BatchNorm2d does.

Reference: models/DispNetS.py:42-140, models/Disp_vgg_BN.py:72-191,
models/Disp_vgg_BN_DORN.py:72-227 (torchvision vgg16_bn cfg "D" supplies the layout only).
"""
import math

import torch
import torch.nn.functional as F

# torchvision vgg16_bn "D" configuration; index i of features.features.<i> follows from it.
VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")
# reference slices features[0:7], [7:14], [14:24], [24:34], [34:44]  (Disp_vgg_BN.py:137-141)
VGG_STAGE_SLICES = ((0, 7), (7, 14), (14, 24), (24, 34), (34, 44))
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def vgg16_bn_layout():
    """[(index, kind, c_in, c_out)] for features.features; kind in {conv,bn,relu,pool}."""
    out, idx, c_in = [], 0, 3
    for v in VGG16_CFG:
        if v == "M":
            out.append((idx, "pool", c_in, c_in)); idx += 1
        else:
            out.append((idx, "conv", c_in, v)); idx += 1
            out.append((idx, "bn", v, v)); idx += 1
            out.append((idx, "relu", v, v)); idx += 1
            c_in = v
    return out


def alpha_beta(datasets):
    # models/Disp_vgg_BN.py:77-82, models/DispNetS.py:47-52
    if datasets == "kitti":
        return 10.0, 0.01
    if datasets == "nyu":
        return 10.0, 0.1
    raise ValueError("undefined datasets %r" % (datasets,))


# ----------------------------------------------------------------------------- state dicts
def _conv_entry(sd, name, c_out, c_in, k, dtype, transposed=False):
    shape = (c_in, c_out, k, k) if transposed else (c_out, c_in, k, k)
    sd[name + ".weight"] = torch.zeros(shape, dtype=dtype)
    sd[name + ".bias"] = torch.zeros(c_out, dtype=dtype)


def disp_vgg_bn_state_dict(dtype=torch.float32, with_classifier=False, dorn_ordinal_c=None):
    """Zero-filled state_dict with the reference's keys/shapes (125 keys incl. classifier)."""
    sd = {}
    for idx, kind, c_in, c_out in vgg16_bn_layout():
        p = "features.features.%d" % idx
        if kind == "conv":
            _conv_entry(sd, p, c_out, c_in, 3, dtype)
        elif kind == "bn":
            sd[p + ".weight"] = torch.ones(c_out, dtype=dtype)
            sd[p + ".bias"] = torch.zeros(c_out, dtype=dtype)
            sd[p + ".running_mean"] = torch.zeros(c_out, dtype=dtype)
            sd[p + ".running_var"] = torch.ones(c_out, dtype=dtype)
            sd[p + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    if with_classifier:
        for i, (o, c) in zip((0, 3, 6), ((4096, 512 * 7 * 7), (4096, 4096), (1000, 4096))):
            sd["features.classifier.%d.weight" % i] = torch.zeros(o, c, dtype=dtype)
            sd["features.classifier.%d.bias" % i] = torch.zeros(o, dtype=dtype)
    for lvl, (ci, co) in zip((4, 3, 2, 1, 0), ((512, 256), (256, 128), (128, 64), (64, 32), (32, 16))):
        _conv_entry(sd, "upconv%d.0" % lvl, co, ci, 4, dtype, transposed=True)
    for lvl, (ci, co) in zip((4, 3, 2, 1, 0), ((768, 256), (384, 128), (193, 64), (97, 32), (17, 16))):
        _conv_entry(sd, "iconv%d.0" % lvl, co, ci, 3, dtype)
    heads = (3, 2, 1) if dorn_ordinal_c else (3, 2, 1, 0)
    for lvl, ci in zip((3, 2, 1, 0), (128, 64, 32, 16)):
        if lvl in heads:
            _conv_entry(sd, "disp%d.0" % lvl, 1, ci, 3, dtype)
    if dorn_ordinal_c:
        _conv_entry(sd, "conv_ord", 2 * dorn_ordinal_c, 16, 1, dtype)
    return sd


DISPNETS_CONV_PLANES = (32, 64, 128, 256, 512, 512, 512)
DISPNETS_UPCONV_PLANES = (512, 512, 256, 128, 64, 32, 16)
DISPNETS_KERNELS = (7, 5, 3, 3, 3, 3, 3)


def dispnets_state_dict(dtype=torch.float32):
    sd, c_in = {}, 3
    for i, (c, k) in enumerate(zip(DISPNETS_CONV_PLANES, DISPNETS_KERNELS), start=1):
        _conv_entry(sd, "conv%d.0" % i, c, c_in, k, dtype)
        _conv_entry(sd, "conv%d.2" % i, c, c, k, dtype)
        c_in = c
    cp, up = DISPNETS_CONV_PLANES, DISPNETS_UPCONV_PLANES
    ins = (cp[6], up[0], up[1], up[2], up[3], up[4], up[5])
    for lvl, ci, co in zip((7, 6, 5, 4, 3, 2, 1), ins, up):
        _conv_entry(sd, "upconv%d.0" % lvl, co, ci, 3, dtype, transposed=True)
    iin = (up[0] + cp[5], up[1] + cp[4], up[2] + cp[3], up[3] + cp[2], 1 + up[4] + cp[1], 1 + up[5] + cp[0], 1 + up[6])
    for lvl, ci, co in zip((7, 6, 5, 4, 3, 2, 1), iin, up):
        _conv_entry(sd, "iconv%d.0" % lvl, co, ci, 3, dtype)
    for lvl, ci in zip((4, 3, 2, 1), (up[3], up[4], up[5], up[6])):
        _conv_entry(sd, "predict_disp%d.0" % lvl, 1, ci, 3, dtype)
    return sd


def xavier_init_(sd, generator=None):
    """init_weights(): xavier_uniform on conv/convT/linear weights, zero bias; BN defaults kept
    (the BN branch is unreachable in the reference, models/Disp_vgg_BN.py:116-120)."""
    for k, t in sd.items():
        if t.dim() >= 2:
            rf = 1
            for s in t.shape[2:]:
                rf *= s
            bound = math.sqrt(6.0 / ((t.shape[0] + t.shape[1]) * rf))
            t.copy_((torch.rand(t.shape, generator=generator, dtype=torch.float64) * 2 - 1).mul_(bound).to(t.dtype))
        elif k.endswith(".bias") and (k[:-4] + "running_mean") not in sd:
            t.zero_()
    return sd


# ------------------------------------------------------------------------------- forwards
def _bn(sd, p, x, training):
    rm, rv = sd[p + ".running_mean"], sd[p + ".running_var"]
    if training:
        sd[p + ".num_batches_tracked"] += 1
    return F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], training, BN_MOMENTUM, BN_EPS)


def vgg_bn_encoder(sd, x, training, prefix="features.features."):
    """Five stage outputs conv1..conv5 (each ends with its 2x2 max-pool)."""
    layout = {idx: (kind, ci, co) for idx, kind, ci, co in vgg16_bn_layout()}
    feats = []
    for lo, hi in VGG_STAGE_SLICES:
        for idx in range(lo, hi):
            kind = layout[idx][0]
            p = prefix + str(idx)
            if kind == "conv":
                x = F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=1, padding=1)
            elif kind == "bn":
                x = _bn(sd, p, x, training)
            elif kind == "relu":
                x = F.relu(x)
            else:
                x = F.max_pool2d(x, 2, 2)
        feats.append(x)
    return feats


def _up_lrelu(sd, name, x, k=4, pad=1, out_pad=0, slope=0.1):
    y = F.conv_transpose2d(x, sd[name + ".0.weight"], sd[name + ".0.bias"], stride=2, padding=pad, output_padding=out_pad)
    return F.leaky_relu(y, slope)


def _iconv_lrelu(sd, name, x, slope=0.1):
    return F.leaky_relu(F.conv2d(x, sd[name + ".0.weight"], sd[name + ".0.bias"], padding=1), slope)


def _head(sd, name, x, alpha, beta):
    return alpha * torch.sigmoid(F.conv2d(x, sd[name + ".0.weight"], sd[name + ".0.bias"], padding=1)) + beta


def _nearest2(x):
    return F.interpolate(x, scale_factor=2, mode="nearest")


def _vgg_decoder_trunk(sd, feats, alpha, beta):
    c1, c2, c3, c4, c5 = feats
    i4 = _iconv_lrelu(sd, "iconv4", torch.cat((_up_lrelu(sd, "upconv4", c5), c4), 1))
    i3 = _iconv_lrelu(sd, "iconv3", torch.cat((_up_lrelu(sd, "upconv3", i4), c3), 1))
    d3 = _head(sd, "disp3", i3, alpha, beta)
    i2 = _iconv_lrelu(sd, "iconv2", torch.cat((_up_lrelu(sd, "upconv2", i3), c2, _nearest2(d3)), 1))
    d2 = _head(sd, "disp2", i2, alpha, beta)
    i1 = _iconv_lrelu(sd, "iconv1", torch.cat((_up_lrelu(sd, "upconv1", i2), c1, _nearest2(d2)), 1))
    d1 = _head(sd, "disp1", i1, alpha, beta)
    i0 = _iconv_lrelu(sd, "iconv0", torch.cat((_up_lrelu(sd, "upconv0", i1), _nearest2(d1)), 1))
    return i0, d1, d2, d3


def disp_vgg_bn(sd, x, training=True, datasets="kitti", only_train_dec=False):
    """models/Disp_vgg_BN.py:136-191.  Returns (disp0..disp3) when training else disp0."""
    alpha, beta = alpha_beta(datasets)
    feats = vgg_bn_encoder(sd, x, training)
    if only_train_dec:
        feats = [f.detach() for f in feats]
    i0, d1, d2, d3 = _vgg_decoder_trunk(sd, feats, alpha, beta)
    d0 = _head(sd, "disp0", i0, alpha, beta)
    return (d0, d1, d2, d3) if training else d0


def ordinal_regression(pre_ord):
    """models/Disp_vgg_BN_DORN.py:196-227: even/odd logit pairs, logits clamped to [1e-8,1e8]
    (acts like a ReLU), 2-way softmax, P(odd) kept; decode = count(P>0.5)."""
    a = pre_ord[:, 0::2].clamp(1e-8, 1e8)
    b = pre_ord[:, 1::2].clamp(1e-8, 1e8)
    ord_c1 = torch.softmax(torch.stack((a, b), 0), 0)[1]
    decode = (ord_c1 > 0.5).sum(1, keepdim=True)
    return decode, ord_c1


def disp_vgg_bn_dorn(sd, x, training=True, datasets="kitti", only_train_dec=False, dropout_mask=None):
    """models/Disp_vgg_BN_DORN.py:139-194.  `dropout_mask` ([N,16,1,1], already scaled by 1/(1-p))
    injects the Dropout2d(0.5) pattern in training mode; None in eval (identity)."""
    alpha, beta = alpha_beta(datasets)
    feats = vgg_bn_encoder(sd, x, training)
    if only_train_dec:
        feats = [f.detach() for f in feats]
    i0, _, _, _ = _vgg_decoder_trunk(sd, feats, alpha, beta)
    if training and dropout_mask is not None:
        i0 = i0 * dropout_mask
    pre = F.conv2d(i0, sd["conv_ord.weight"], sd["conv_ord.bias"])
    return ordinal_regression(pre)


def _crop_like(x, ref):
    return x[:, :, : ref.shape[2], : ref.shape[3]]


def dispnets(sd, x, training=True, datasets="kitti"):
    """models/DispNetS.py:93-140.  Returns (disp1..disp4) when training else disp1."""
    alpha, beta = alpha_beta(datasets)
    enc, h = [], x
    for i, k in enumerate(DISPNETS_KERNELS, start=1):
        p = (k - 1) // 2
        h = F.relu(F.conv2d(h, sd["conv%d.0.weight" % i], sd["conv%d.0.bias" % i], stride=2, padding=p))
        h = F.relu(F.conv2d(h, sd["conv%d.2.weight" % i], sd["conv%d.2.bias" % i], stride=1, padding=p))
        enc.append(h)

    def up(name, t):
        return F.relu(F.conv_transpose2d(t, sd[name + ".0.weight"], sd[name + ".0.bias"], stride=2, padding=1, output_padding=1))

    def ic(name, t):
        return F.relu(F.conv2d(t, sd[name + ".0.weight"], sd[name + ".0.bias"], padding=1))

    def bil2(t):
        return F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)

    c1, c2, c3, c4, c5, c6, c7 = enc
    i7 = ic("iconv7", torch.cat((_crop_like(up("upconv7", c7), c6), c6), 1))
    i6 = ic("iconv6", torch.cat((_crop_like(up("upconv6", i7), c5), c5), 1))
    i5 = ic("iconv5", torch.cat((_crop_like(up("upconv5", i6), c4), c4), 1))
    i4 = ic("iconv4", torch.cat((_crop_like(up("upconv4", i5), c3), c3), 1))
    d4 = _head(sd, "predict_disp4", i4, alpha, beta)
    i3 = ic("iconv3", torch.cat((_crop_like(up("upconv3", i4), c2), c2, _crop_like(bil2(d4), c2)), 1))
    d3 = _head(sd, "predict_disp3", i3, alpha, beta)
    i2 = ic("iconv2", torch.cat((_crop_like(up("upconv2", i3), c1), c1, _crop_like(bil2(d3), c1)), 1))
    d2 = _head(sd, "predict_disp2", i2, alpha, beta)
    i1 = ic("iconv1", torch.cat((_crop_like(up("upconv1", i2), x), _crop_like(bil2(d2), x)), 1))
    d1 = _head(sd, "predict_disp1", i1, alpha, beta)
    return (d1, d2, d3, d4) if training else d1
