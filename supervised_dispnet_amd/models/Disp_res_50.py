"""Drop-in mirror of the reference's models/Disp_res_50.py (ResNet-50 encoder + DispNet decoder) on the HIP engine.

Same constructor signature, `init_weights(use_pretrained_weights)`, state_dict keys (conv1, bn1, layer{1..4}.{i}.{conv,bn}{1,2,3},
layer*.0.downsample.{0,1}, upconv{5..1}.0, iconv{5..1}.0, predict_disp{4..1}.0) and forward contract (disp1..disp4 in training
mode, disp1 in eval mode) -- reference models/Disp_res_50.py:49-247.

Reference quirk kept: `bn1` is computed and its output DISCARDED (`relu1 = self.relu(conv1)`, :141-145): its affine parameters
receive no gradient, but its running statistics are updated in training mode.
"""
import torch.nn as nn

from .. import engine
from .._lib import ACT_LEAKY, ACT_RELU, ACT_SIGMOID_AFFINE
from ._common import run_net, xavier_init_like_reference


def predict_disp(in_planes):
    return nn.Sequential(nn.Conv2d(in_planes, 1, kernel_size=3, padding=1), nn.Sigmoid())


def conv(in_planes, out_planes):
    return nn.Sequential(nn.Conv2d(in_planes, out_planes, kernel_size=3, padding=1), nn.LeakyReLU(0.1))


def upconv(in_planes, out_planes):
    return nn.Sequential(nn.ConvTranspose2d(in_planes, out_planes, kernel_size=3, stride=2, padding=1, output_padding=1), nn.LeakyReLU(0.1))


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def conv1x1(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


class Bottleneck(nn.Module):
    """Parameter container with the reference's layout (models/Disp_res_50.py:212-247); executed by run_bottleneck."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(Bottleneck, self).__init__()
        self.conv1 = conv1x1(inplanes, planes)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = conv3x3(planes, planes, stride)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = conv1x1(planes, planes * 4)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


class BasicBlock(nn.Module):
    """torchvision BasicBlock layout (ResNet-18/34): conv3x3-bn-relu-conv3x3-bn (+ downsample) + add + relu."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(BasicBlock, self).__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


def block_runtime(blk):
    """engine.ConvLayer companions of one residual block (cached on the module)."""
    rt = getattr(blk, "_dn_rt", None)
    if rt is None:
        rt = {"conv1": engine.ConvLayer(blk.conv1), "conv2": engine.ConvLayer(blk.conv2)}
        if hasattr(blk, "conv3"):
            rt["conv3"] = engine.ConvLayer(blk.conv3)
        if blk.downsample is not None:
            rt["ds"] = engine.ConvLayer(blk.downsample[0])
        object.__setattr__(blk, "_dn_rt", rt)
    return rt


def run_residual_block(tape, sink, x, blk, training):
    """x: plain activation -> plain activation.  Bottleneck: 1x1-BN-ReLU, 3x3(stride)-BN-ReLU, 1x1-BN, (+ downsample 1x1-BN),
    add, ReLU (reference :229-247); BasicBlock likewise with two 3x3.  BN-apply+ReLU of the inner layers is fused into the
    next conv's loader; the tail (bn + residual + relu) is one pass."""
    rt = block_runtime(blk)
    P = engine.Piece
    y = engine.block_conv_bn(tape, sink, P(x), rt["conv1"], blk.bn1, training)
    if "conv3" in rt:
        y = engine.block_conv_bn(tape, sink, P(y), rt["conv2"], blk.bn2, training)
        y = engine.block_conv_bn(tape, sink, P(y), rt["conv3"], blk.bn3, training, relu=False)
    else:
        y = engine.block_conv_bn(tape, sink, P(y), rt["conv2"], blk.bn2, training, relu=False)
    r = x
    if blk.downsample is not None:
        r = engine.block_conv_bn(tape, sink, P(x), rt["ds"], blk.downsample[1], training, relu=False)
    return engine.block_bn_add_relu(tape, y, r)


def residual_block_params_backward_order(blk):
    """Parameters of one block in the order backward produces their gradients."""
    out = []
    if blk.downsample is not None:
        out += [blk.downsample[1].weight, blk.downsample[1].bias, blk.downsample[0].weight]
    names = [("bn3", "conv3"), ("bn2", "conv2"), ("bn1", "conv1")] if hasattr(blk, "conv3") else [("bn2", "conv2"), ("bn1", "conv1")]
    for bn, cv in names:
        out += [getattr(blk, bn).weight, getattr(blk, bn).bias, getattr(blk, cv).weight]
    return out


class Disp_res_50(nn.Module):
    # what Disp_res_18 (reference models/Disp_res_18.py:50-135: BasicBlock, expansion written out as 1) changes
    _block, _expansion, _counts = Bottleneck, 4, (3, 4, 6, 3)
    _pretrained_url = 'https://download.pytorch.org/models/resnet50-19c8e357.pth'

    def __init__(self, datasets='kitti'):
        super(Disp_res_50, self).__init__()
        if datasets == 'kitti':
            self.alpha, self.beta = 10, 0.01
        elif datasets == 'nyu':
            self.alpha, self.beta = 10, 0.1
        else:
            raise ValueError("undefined datasets %r" % (datasets,))
        self.only_train_dec = False
        self.inplanes = 64
        conv_planes = [64, 64, 128, 256, 512]
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.pool1 = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        ex, cnt = self._expansion, self._counts
        self.layer1 = self.resblock(conv_planes[1], cnt[0])
        self.layer2 = self.resblock(conv_planes[2], cnt[1], stride=2)
        self.layer3 = self.resblock(conv_planes[3], cnt[2], stride=2)
        self.layer4 = self.resblock(conv_planes[4], cnt[3], stride=2)
        upconv_planes = [512, 256, 128, 64, 32, 16]
        self.upconv5 = upconv(conv_planes[4] * ex, upconv_planes[1])
        self.upconv4 = upconv(upconv_planes[1], upconv_planes[2])
        self.upconv3 = upconv(upconv_planes[2], upconv_planes[3])
        self.upconv2 = upconv(upconv_planes[3], upconv_planes[4])
        self.upconv1 = upconv(upconv_planes[4], upconv_planes[5])
        self.iconv5 = conv(upconv_planes[1] + conv_planes[3] * ex, upconv_planes[1])
        self.iconv4 = conv(upconv_planes[2] + conv_planes[2] * ex, upconv_planes[2])
        self.iconv3 = conv(1 + upconv_planes[3] + conv_planes[1] * ex, upconv_planes[3])
        self.iconv2 = conv(1 + upconv_planes[4] + conv_planes[0], upconv_planes[4])
        self.iconv1 = conv(1 + upconv_planes[5], upconv_planes[5])
        self.predict_disp4 = predict_disp(upconv_planes[2])
        self.predict_disp3 = predict_disp(upconv_planes[3])
        self.predict_disp2 = predict_disp(upconv_planes[4])
        self.predict_disp1 = predict_disp(upconv_planes[5])
        self._rt = None

    def resblock(self, planes, num_blocks, stride=1):
        downsample, ex = None, self._expansion
        if stride != 1 or self.inplanes != planes * ex:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * ex, stride), nn.BatchNorm2d(planes * ex))
        layers = [self._block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * ex
        for _ in range(1, num_blocks):
            layers.append(self._block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def init_weights(self, use_pretrained_weights=False):
        xavier_init_like_reference(self)
        if use_pretrained_weights:
            import torch.utils.model_zoo as model_zoo
            print("loading pretrained weights downloaded from pytorch.org")
            self.load_res_params(model_zoo.load_url(self._pretrained_url))
        else:
            print("do not load pretrained weights for the monocular model")

    def load_res_params(self, params):
        model_dict = self.state_dict()
        model_dict.update({k: v for k, v in params.items() if k in model_dict})
        self.load_state_dict(model_dict)

    def forward(self, x):
        outs = run_net(self, x)
        return outs if self.training else outs[0]

    # ------------------------------------------------------------------ engine side
    def _hot_parameters(self):
        """Everything except bn1's affine pair, which the reference's forward never uses (:141-145)."""
        skip = {id(self.bn1.weight), id(self.bn1.bias)}
        return [p for p in self.parameters() if id(p) not in skip]

    def _grad_production_order(self):
        order = []
        for name in ("predict_disp1", "iconv1", "upconv1", "predict_disp2", "iconv2", "upconv2", "predict_disp3", "iconv3", "upconv3",
                     "predict_disp4", "iconv4", "upconv4", "iconv5", "upconv5"):
            m = getattr(self, name)[0]
            order += [m.bias, m.weight]
        for layer in (self.layer4, self.layer3, self.layer2, self.layer1):
            for blk in reversed(list(layer)):
                order += residual_block_params_backward_order(blk)
        order.append(self.conv1.weight)
        return order

    def _runtime(self):
        if self._rt is None:
            rt = {"conv1": engine.ConvLayer(self.conv1)}
            for i in range(1, 6):
                rt["upconv%d" % i] = engine.ConvLayer(getattr(self, "upconv%d" % i)[0], transposed=True)
                rt["iconv%d" % i] = engine.ConvLayer(getattr(self, "iconv%d" % i)[0])
            for i in range(1, 5):
                rt["predict_disp%d" % i] = engine.ConvLayer(getattr(self, "predict_disp%d" % i)[0])
            self._rt = rt
        return self._rt

    def _hip_forward(self, tape, sink, x):
        rt = self._runtime()
        P = engine.Piece
        a, b = float(self.alpha), float(self.beta)
        training = self.training
        relu1 = engine.block_conv_act(tape, sink, [P(x)], rt["conv1"], ACT_RELU, stat_bn=self.bn1 if training else None)
        pool1 = engine.block_maxpool3s2(tape, relu1)
        feats, cur = [], pool1
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                cur = run_residual_block(tape, sink, cur, blk, training)
            feats.append(cur)
        conv2, conv3, conv4, conv5 = feats
        if self.only_train_dec:                       # .detach() of relu1, pool1, conv2..conv5 (reference :153-159)
            for act in (relu1, pool1, conv2, conv3, conv4, conv5):
                act.needs_grad = False
        lrelu = lambda name, pieces: engine.block_conv_act(tape, sink, pieces, rt[name], ACT_LEAKY, 0.1)
        head = lambda name, act: engine.block_conv_act(tape, sink, [P(act)], rt[name], ACT_SIGMOID_AFFINE, a, b)
        i5 = lrelu("iconv5", [P(lrelu("upconv5", [P(conv5)])), P(conv4)])
        i4 = lrelu("iconv4", [P(lrelu("upconv4", [P(i5)])), P(conv3)])
        d4 = head("predict_disp4", i4)
        i3 = lrelu("iconv3", [P(lrelu("upconv3", [P(i4)])), P(conv2), P(d4, up=True)])
        d3 = head("predict_disp3", i3)
        i2 = lrelu("iconv2", [P(lrelu("upconv2", [P(i3)])), P(relu1), P(d3, up=True)])
        d2 = head("predict_disp2", i2)
        i1 = lrelu("iconv1", [P(lrelu("upconv1", [P(i2)])), P(d2, up=True)])
        d1 = head("predict_disp1", i1)
        return [d1, d2, d3, d4]
