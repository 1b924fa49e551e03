"""Drop-in mirrors of the reference's models/Disp_res.py and models/Disp_res_101.py: ResNet-50 / ResNet-101 bottleneck encoder
with the SIX-level decoder (upconv6..1, iconv6..1), `crop_like` after every up-convolution and bilinear x2 disparity upsampling.

Reference: models/Disp_res.py:59-208 (LeakyReLU(0.1) decoder, layer3 = 6 blocks), models/Disp_res_101.py:43-196 (ReLU decoder,
layer3 = 23 blocks).  Both forward passes are the same text; what it does, quirks included, is kept:
  * bn1 is evaluated and its output discarded (`relu1 = self.relu(conv1)`): running statistics update, no gradient to its affine pair;
  * skip2 = pool1 sits at H/4 while upconv3's result is at H/2, so `crop_like(upconv3(iconv4), skip2)` keeps the TOP-LEFT QUARTER of
    the up-convolution (and of the bilinear-upsampled disp4): iconv3 / disp3 run at H/4, and the outputs are disp1 @H, disp2 @H/2,
    disp3 @H/4, disp4 @H/4.  Here the transposed convolution only computes the cropped window (out_hw) -- same values, no waste.
State_dict keys: conv1, bn1, layer{1..4}.*, upconv{6..1}.0, iconv{6..1}.0, predict_disp{4..1}.0.
"""
import torch.nn as nn

from .. import engine
from .._lib import ACT_LEAKY, ACT_RELU, ACT_SIGMOID_AFFINE
from ._common import run_net, xavier_init_like_reference
from .Disp_res_50 import Bottleneck, conv1x1, predict_disp, residual_block_params_backward_order, run_residual_block


def _conv(in_planes, out_planes, leaky):
    return nn.Sequential(nn.Conv2d(in_planes, out_planes, kernel_size=3, padding=1), nn.LeakyReLU(0.1) if leaky else nn.ReLU(inplace=True))


def _upconv(in_planes, out_planes, leaky):
    return nn.Sequential(nn.ConvTranspose2d(in_planes, out_planes, kernel_size=3, stride=2, padding=1, output_padding=1),
                         nn.LeakyReLU(0.1) if leaky else nn.ReLU(inplace=True))


class Disp_res(nn.Module):
    _layer3_blocks = 6
    _leaky = True                 # models/Disp_res.py:17-39: conv()/upconv() default to leaky=True
    _pretrained_url = 'https://download.pytorch.org/models/resnet50-19c8e357.pth'

    def __init__(self, datasets='kitti'):
        super(Disp_res, self).__init__()
        if datasets == 'kitti':
            self.alpha, self.beta = 10, 0.01
        elif datasets == 'nyu':
            self.alpha, self.beta = 10, 0.1
        else:
            raise ValueError("undefined datasets %r" % (datasets,))
        self.only_train_dec = False
        self.inplanes = 64
        cp = [64, 64, 128, 256, 512]
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.pool1 = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self.resblock(cp[1], 3)
        self.layer2 = self.resblock(cp[2], 4, stride=2)
        self.layer3 = self.resblock(cp[3], self._layer3_blocks, stride=2)
        self.layer4 = self.resblock(cp[4], 3, stride=2)
        up, lk = [512, 256, 128, 64, 32, 16], self._leaky
        self.upconv6 = _upconv(cp[4] * 4, up[0], lk)
        self.upconv5 = _upconv(up[0], up[1], lk)
        self.upconv4 = _upconv(up[1], up[2], lk)
        self.upconv3 = _upconv(up[2], up[3], lk)
        self.upconv2 = _upconv(up[3], up[4], lk)
        self.upconv1 = _upconv(up[4], up[5], lk)
        self.iconv6 = _conv(up[0] + cp[3] * 4, up[0], lk)
        self.iconv5 = _conv(up[1] + cp[2] * 4, up[1], lk)
        self.iconv4 = _conv(up[2] + cp[1] * 4, up[2], lk)
        self.iconv3 = _conv(1 + up[3] + cp[1], up[3], lk)
        self.iconv2 = _conv(1 + up[4] + cp[0], up[4], lk)
        self.iconv1 = _conv(1 + up[5], up[5], lk)
        self.predict_disp4 = predict_disp(up[2])
        self.predict_disp3 = predict_disp(up[3])
        self.predict_disp2 = predict_disp(up[4])
        self.predict_disp1 = predict_disp(up[5])
        self._rt = None

    def resblock(self, planes, num_blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * 4, stride), nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        for _ in range(1, num_blocks):
            layers.append(Bottleneck(self.inplanes, planes))
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
        skip = {id(self.bn1.weight), id(self.bn1.bias)}
        return [p for p in self.parameters() if id(p) not in skip]

    def _grad_production_order(self):
        order = []
        for name in ("predict_disp1", "iconv1", "upconv1", "predict_disp2", "iconv2", "upconv2", "predict_disp3", "iconv3", "upconv3",
                     "predict_disp4", "iconv4", "upconv4", "iconv5", "upconv5", "iconv6", "upconv6"):
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
            for i in range(1, 7):
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
        dact, slope = (ACT_LEAKY, 0.1) if self._leaky else (ACT_RELU, 0.0)
        relu1 = engine.block_conv_act(tape, sink, [P(x)], rt["conv1"], ACT_RELU, stat_bn=self.bn1 if training else None)
        pool1 = engine.block_maxpool3s2(tape, relu1)
        feats, cur = [], pool1
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                cur = run_residual_block(tape, sink, cur, blk, training)
            feats.append(cur)
        conv2, conv3, conv4, conv5 = feats
        if self.only_train_dec:                       # reference models/Disp_res.py:150-156
            for act in (relu1, pool1, conv2, conv3, conv4, conv5):
                act.needs_grad = False
        hw = lambda act: (act.H, act.W)
        dec = lambda name, pieces, out_hw=None: engine.block_conv_act(tape, sink, pieces, rt[name], dact, slope, out_hw=out_hw)
        head = lambda name, act: engine.block_conv_act(tape, sink, [P(act)], rt[name], ACT_SIGMOID_AFFINE, a, b)
        bil = lambda d, ref_hw: engine.block_bilinear_up2(tape, d, ref_hw)
        skip1, skip2, skip3, skip4, skip5 = relu1, pool1, conv2, conv3, conv4
        i6 = dec("iconv6", [P(dec("upconv6", [P(conv5)], hw(skip5))), P(skip5)])
        i5 = dec("iconv5", [P(dec("upconv5", [P(i6)], hw(skip4))), P(skip4)])
        i4 = dec("iconv4", [P(dec("upconv4", [P(i5)], hw(skip3))), P(skip3)])
        d4 = head("predict_disp4", i4)
        i3 = dec("iconv3", [P(dec("upconv3", [P(i4)], hw(skip2))), P(skip2), P(bil(d4, hw(skip2)))])
        d3 = head("predict_disp3", i3)
        i2 = dec("iconv2", [P(dec("upconv2", [P(i3)], hw(skip1))), P(skip1), P(bil(d3, hw(skip1)))])
        d2 = head("predict_disp2", i2)
        i1 = dec("iconv1", [P(dec("upconv1", [P(i2)], hw(x))), P(bil(d2, hw(x)))])
        d1 = head("predict_disp1", i1)
        return [d1, d2, d3, d4]
