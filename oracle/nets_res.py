"""Oracle (torch-CPU) restatement of the reference's ResNet-backbone / monodepth2-style nets and PoseExpNet.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Like oracle/nets.py the nets are functions of a state_dict keyed exactly
like the reference modules' state_dict(); BatchNorm running statistics are updated in place in training mode.

Reference: models/Disp_res_50.py:49-247, networks/resnet_encoder.py:60-98, networks/vgg_encoder.py:60-87,
networks/depth_decoder.py:17-70 + layers.py:106-136,193-196, models/monodepth2.py:8-16, models/PoseExpNet.py:20-95.
"""
import torch
import torch.nn.functional as F

from .nets import _bn, _head, _nearest2, alpha_beta, vgg_bn_encoder

RESNET_LAYERS = {18: ("basic", (2, 2, 2, 2)), 34: ("basic", (3, 4, 6, 3)), 50: ("bottleneck", (3, 4, 6, 3))}


# --------------------------------------------------------------------------------------------------- residual blocks
def bottleneck(sd, p, x, stride, training):
    """models/Disp_res_50.py:229-247: 1x1-BN-ReLU, 3x3(stride)-BN-ReLU, 1x1-BN, optional downsample (1x1 stride + BN), add, ReLU."""
    out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"]), training))
    out = F.relu(_bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], stride=stride, padding=1), training))
    out = _bn(sd, p + ".bn3", F.conv2d(out, sd[p + ".conv3.weight"]), training)
    identity = x
    if (p + ".downsample.0.weight") in sd:
        identity = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), training)
    return F.relu(out + identity)


def basic_block(sd, p, x, stride, training):
    """torchvision BasicBlock (used by networks/resnet_encoder.py for 18/34 layers)."""
    out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], stride=stride, padding=1), training))
    out = _bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], padding=1), training)
    identity = x
    if (p + ".downsample.0.weight") in sd:
        identity = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), training)
    return F.relu(out + identity)


def res_layers(sd, x, prefix, kind, counts, training):
    """layer1..layer4; the first block of layer2..4 has stride 2.  Returns the four stage outputs."""
    block = bottleneck if kind == "bottleneck" else basic_block
    feats = []
    for li, n in enumerate(counts, start=1):
        for bi in range(n):
            stride = 2 if (li > 1 and bi == 0) else 1
            x = block(sd, "%slayer%d.%d" % (prefix, li, bi), x, stride, training)
        feats.append(x)
    return feats


# ------------------------------------------------------------------------------------------------------ Disp_res_50
def _up3_lrelu(sd, name, x):
    # upconv(): ConvTranspose2d(k3, s2, p1, output_padding 1) + LeakyReLU(0.1)   (models/Disp_res_50.py:30-41)
    return F.leaky_relu(F.conv_transpose2d(x, sd[name + ".0.weight"], sd[name + ".0.bias"], stride=2, padding=1, output_padding=1), 0.1)


def _conv3_lrelu(sd, name, x):
    return F.leaky_relu(F.conv2d(x, sd[name + ".0.weight"], sd[name + ".0.bias"], padding=1), 0.1)


def disp_res_50(sd, x, training=True, datasets="kitti", only_train_dec=False):
    """models/Disp_res_50.py:138-210.  QUIRK (:141-145): bn1 is evaluated (running stats update) and its output discarded;
    relu1 = relu(conv1)."""
    alpha, beta = alpha_beta(datasets)
    conv1 = F.conv2d(x, sd["conv1.weight"], stride=2, padding=3)
    _bn(sd, "bn1", conv1, training)                       # result discarded
    relu1 = F.relu(conv1)
    pool1 = F.max_pool2d(relu1, 3, 2, 1)
    conv2, conv3, conv4, conv5 = res_layers(sd, pool1, "", "bottleneck", (3, 4, 6, 3), training)
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


# -------------------------------------------------------------------------------------------- monodepth2-style pieces
def resnet_encoder(sd, x, num_layers=18, training=True, prefix="encoder."):
    """networks/resnet_encoder.py:87-98: (x-0.45)/0.225; conv1-bn1-relu; maxpool; layer1..4 -> 5 feature maps."""
    kind, counts = RESNET_LAYERS[num_layers]
    x = (x - 0.45) / 0.225
    f0 = F.relu(_bn(sd, prefix + "bn1", F.conv2d(x, sd[prefix + "conv1.weight"], stride=2, padding=3), training))
    return [f0] + res_layers(sd, F.max_pool2d(f0, 3, 2, 1), prefix, kind, counts, training)


def vgg_encoder(sd, x, training=True, prefix="encoder.features."):
    """networks/vgg_encoder.py:78-87."""
    return vgg_bn_encoder(sd, (x - 0.45) / 0.225, training, prefix=prefix)


def _conv3x3_refl(sd, p, x):
    # layers.Conv3x3: ReflectionPad2d(1) + Conv2d(3)   (layers.py:124-136)
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), sd[p + ".conv.weight"], sd[p + ".conv.bias"])


def depth_decoder(sd, feats, training=True, prefix="", scales=(0, 1, 2, 3), use_skips=True):
    """networks/depth_decoder.py:50-70.  `decoder.<j>` indices follow the OrderedDict insertion order (:29-47):
    (upconv,4,0),(upconv,4,1),(upconv,3,0),...,(upconv,0,1),(dispconv,0..3)."""
    idx = {}
    j = 0
    for i in range(4, -1, -1):
        idx[("upconv", i, 0)] = j
        idx[("upconv", i, 1)] = j + 1
        j += 2
    for s in scales:
        idx[("dispconv", s)] = j
        j += 1
    name = lambda key: "%sdecoder.%d" % (prefix, idx[key])
    x = feats[-1]
    outs = {}
    for i in range(4, -1, -1):
        x = F.elu(_conv3x3_refl(sd, name(("upconv", i, 0)) + ".conv", x))
        x = [_nearest2(x)]
        if use_skips and i > 0:
            x += [feats[i - 1]]
        x = F.elu(_conv3x3_refl(sd, name(("upconv", i, 1)) + ".conv", torch.cat(x, 1)))
        if i in scales:
            outs[i] = 0.01 + 9.99 * torch.sigmoid(_conv3x3_refl(sd, name(("dispconv", i)), x))
    return (outs[0], outs[1], outs[2], outs[3]) if training else outs[0]


def monodepth2(sd, x, encoder="vgg", training=True):
    """models/monodepth2.py:13-16 with state_dict keys `encoder.*` / `decoder.*`."""
    if encoder == "vgg":
        feats = vgg_encoder(sd, x, training, prefix="encoder.encoder.features.")
    else:
        feats = resnet_encoder(sd, x, int(encoder), training, prefix="encoder.encoder.")
    return depth_decoder(sd, feats, training, prefix="decoder.")


# --------------------------------------------------------------------------------------------------------- PoseExpNet
def pose_exp_net(sd, target, refs, output_exp=False, training=True):
    """models/PoseExpNet.py:58-95."""
    nb = len(refs)
    x = torch.cat([target] + list(refs), 1)
    size_in = x.shape[2:]
    enc = []
    for i, k in zip(range(1, 8), (7, 5, 3, 3, 3, 3, 3)):
        x = F.relu(F.conv2d(x, sd["conv%d.0.weight" % i], sd["conv%d.0.bias" % i], stride=2, padding=(k - 1) // 2))
        enc.append(x)
    pose = F.conv2d(enc[6], sd["pose_pred.weight"], sd["pose_pred.bias"]).mean(3).mean(2)
    pose = 0.01 * pose.view(pose.size(0), nb, 6)
    masks = [None, None, None, None]
    if output_exp:
        targets = [enc[3].shape[2:], enc[2].shape[2:], enc[1].shape[2:], enc[0].shape[2:], size_in]
        cur, ups = enc[4], {}
        for i, hw in zip((5, 4, 3, 2, 1), targets):
            cur = F.relu(F.conv_transpose2d(cur, sd["upconv%d.0.weight" % i], sd["upconv%d.0.bias" % i], stride=2, padding=1))
            cur = cur[:, :, :hw[0], :hw[1]]
            ups[i] = cur
        masks = [torch.sigmoid(F.conv2d(ups[i], sd["predict_mask%d.weight" % i], sd["predict_mask%d.bias" % i], padding=1)) for i in (1, 2, 3, 4)]
    if training:
        return masks, pose
    return masks[0], pose
