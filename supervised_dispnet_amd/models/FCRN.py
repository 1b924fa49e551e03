"""Drop-in mirror of the reference's models/FCRN.py (ResNet-50 + four up-projection blocks, Laina et al.) on the HIP engine
(SURVEY.md section 8 f-4).

Same constructor signature, `init_weights(use_pretrained_weights)`, state_dict keys (conv1, bn1, layer{1..4}.*, conv2, bn2,
up{1..4}.{conv1_1..conv1_4, conv2_1..conv2_4, bn1_1, bn1_2, conv3, bn2}, conv3) and forward contract ([depth-like map at the input
size] in training mode, the tensor in eval mode) -- reference models/FCRN.py:52-124 (UpProject), :126-260.

What the up-projection is here: the reference pads by hand (F.pad) in front of eight un-padded convolutions and interleaves their
results with stack / permute / view; phase (a, b) of the interleaved map being conv*_(2a+b+1), a branch IS a ConvTranspose2d(6x6,
stride 2, padding 2) on a weight assembled from its four convolutions, and the engine runs it as one four-phase transposed convolution
(no zero insertion) plus the four per-phase biases (engine.block_upproject).  The reference keeps the "author's
interleaving padding" (:73-78, 82-88) -- so does this.
"""
import math

import torch
import torch.nn as nn

from .. import engine
from .._lib import ACT_SIGMOID_AFFINE
from ._common import run_net
from .Disp_res_50 import Bottleneck, conv1x1, run_residual_block, residual_block_params_backward_order


class UpProject(nn.Module):
    """Parameter container with the reference's layout (models/FCRN.py:54-72); executed by engine.block_upproject."""

    def __init__(self, in_channels, out_channels):
        super(UpProject, self).__init__()
        self.conv1_1 = nn.Conv2d(in_channels, out_channels, 3)
        self.conv1_2 = nn.Conv2d(in_channels, out_channels, (2, 3))
        self.conv1_3 = nn.Conv2d(in_channels, out_channels, (3, 2))
        self.conv1_4 = nn.Conv2d(in_channels, out_channels, 2)
        self.conv2_1 = nn.Conv2d(in_channels, out_channels, 3)
        self.conv2_2 = nn.Conv2d(in_channels, out_channels, (2, 3))
        self.conv2_3 = nn.Conv2d(in_channels, out_channels, (3, 2))
        self.conv2_4 = nn.Conv2d(in_channels, out_channels, 2)
        self.bn1_1 = nn.BatchNorm2d(out_channels)
        self.bn1_2 = nn.BatchNorm2d(out_channels)
        self.relu = nn.ReLU(inplace=True)
        self.conv3 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.bn2 = nn.BatchNorm2d(out_channels)


def upproject_runtime(up):
    rt = getattr(up, "_dn_rt", None)
    if rt is None:
        rt = {"conv3": engine.ConvLayer(up.conv3)}
        for br in (1, 2):               # a branch's four convolutions = one ConvTranspose2d(6x6, stride 2, padding 2) on a composite weight
            rt["t%d" % br] = engine.ConvLayer(engine._CompositeConvT([getattr(up, "conv%d_%d" % (br, k)) for k in (1, 2, 3, 4)]), transposed=True)
        object.__setattr__(up, "_dn_rt", rt)
    return rt


def upproject_params_backward_order(up):
    out = [up.bn2.weight, up.bn2.bias, up.conv3.bias, up.conv3.weight]
    for br, bn in ((1, up.bn1_1), (2, up.bn1_2)):
        out += [bn.weight, bn.bias]
        for k in (1, 2, 3, 4):
            m = getattr(up, "conv%d_%d" % (br, k))
            out += [m.bias, m.weight]
    return out


class FCRN(nn.Module):
    def __init__(self, datasets='kitti'):
        super(FCRN, self).__init__()
        if datasets == 'kitti':
            self.alpha, self.beta = 10, 0.01
        elif datasets == 'nyu':
            self.alpha, self.beta = 10, 0.1
        else:
            raise ValueError("undefined datasets %r" % (datasets,))
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(Bottleneck, 64, 3, stride=1)
        self.layer2 = self._make_layer(Bottleneck, 128, 4, stride=2)
        self.layer3 = self._make_layer(Bottleneck, 256, 6, stride=2)
        self.layer4 = self._make_layer(Bottleneck, 512, 3, stride=2)
        self.conv2 = nn.Conv2d(2048, 1024, kernel_size=1, bias=False)
        self.bn2 = nn.BatchNorm2d(1024)
        self.up1 = UpProject(1024, 512)
        self.up2 = UpProject(512, 256)
        self.up3 = UpProject(256, 128)
        self.up4 = UpProject(128, 64)
        self.drop = nn.Dropout2d()
        self.conv3 = nn.Conv2d(64, 1, 3, padding=1)
        self.Sigmoid = nn.Sigmoid()
        self._rt = None
        self._dropout_mask = None                           # test hook: fixed [N,64] keep/scale mask instead of RNG

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride), nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def init_weights(self, use_pretrained_weights=False):
        """reference :165-180: Conv2d weights ~ N(0, sqrt(2 / (k_h * k_w * out_channels))), BatchNorm gamma 1 / beta 0."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        if use_pretrained_weights:
            import torchvision.models as tvm
            print("loading pretrained weights downloaded from pytorch.org")
            self.load_state_dict(tvm.resnet50(pretrained=True).state_dict(), strict=False)
        else:
            print("do not load pretrained weights for the monocular model")

    def load_res_params(self, params):
        model_dict = self.state_dict()
        model_dict.update({k: v for k, v in params.items() if k in model_dict})
        self.load_state_dict(model_dict)

    def forward(self, x):
        outs = run_net(self, x)
        return [outs[0]] if self.training else outs[0]

    # ------------------------------------------------------------------ engine side
    def _hot_parameters(self):
        return list(self.parameters())

    def _grad_production_order(self):
        order = [self.conv3.bias, self.conv3.weight]
        for up in (self.up4, self.up3, self.up2, self.up1):
            order += upproject_params_backward_order(up)
        order += [self.bn2.weight, self.bn2.bias, self.conv2.weight]
        for layer in (self.layer4, self.layer3, self.layer2, self.layer1):
            for blk in reversed(list(layer)):
                order += residual_block_params_backward_order(blk)
        order += [self.bn1.weight, self.bn1.bias, self.conv1.weight]
        return order

    def _runtime(self):
        if self._rt is None:
            self._rt = {"conv1": engine.ConvLayer(self.conv1), "conv2": engine.ConvLayer(self.conv2), "conv3": engine.ConvLayer(self.conv3)}
        return self._rt

    def _hip_forward(self, tape, sink, x):
        rt = self._runtime()
        P = engine.Piece
        training = self.training
        cur = engine.block_conv_bn(tape, sink, P(x), rt["conv1"], self.bn1, training)
        cur = engine.block_maxpool3s2(tape, engine.block_bn_relu(tape, cur))
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                cur = run_residual_block(tape, sink, cur, blk, training)
        cur = engine.block_bn_plain(tape, engine.block_conv_bn(tape, sink, P(cur), rt["conv2"], self.bn2, training, relu=False))
        for up in (self.up1, self.up2, self.up3, self.up4):
            cur = engine.block_upproject(tape, sink, cur, up, upproject_runtime(up), training)
        if training and self.drop.p > 0:
            mask = self._dropout_mask
            if mask is None:
                keep = 1.0 - self.drop.p
                mask = torch.bernoulli(torch.full((cur.N, cur.C), keep, dtype=torch.float32, device=cur.t.device)) / keep
            cur = engine.block_channel_scale(tape, cur, mask.to(cur.t.device).contiguous().float())
        head = engine.block_conv_act(tape, sink, [P(cur)], rt["conv3"], ACT_SIGMOID_AFFINE, float(self.alpha), float(self.beta))
        return [engine.block_resize_bilinear(tape, head, (x.H, x.W), align_corners=True)]
