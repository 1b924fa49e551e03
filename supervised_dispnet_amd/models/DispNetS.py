"""Drop-in mirror of the reference's models/DispNetS.py (SfmLearner DispNetS) on the HIP engine.

Same constructor signature, `init_weights`, state_dict keys (conv1.0 ... predict_disp1.0), forward contract:
(disp1, disp2, disp3, disp4) in training mode, disp1 in eval mode -- reference models/DispNetS.py:42-140.
"""
import torch.nn as nn

from .. import engine
from .._lib import ACT_RELU, ACT_SIGMOID_AFFINE
from ._common import run_net, xavier_init_like_reference


def _downsample_conv(c_in, c_out, k=3):
    # reference :7-13
    return nn.Sequential(nn.Conv2d(c_in, c_out, kernel_size=k, stride=2, padding=(k - 1) // 2), nn.ReLU(inplace=True),
                         nn.Conv2d(c_out, c_out, kernel_size=k, padding=(k - 1) // 2), nn.ReLU(inplace=True))


def _predict_disp(c_in):
    return nn.Sequential(nn.Conv2d(c_in, 1, kernel_size=3, padding=1), nn.Sigmoid())


def _conv(c_in, c_out):
    return nn.Sequential(nn.Conv2d(c_in, c_out, kernel_size=3, padding=1), nn.ReLU(inplace=True))


def _upconv(c_in, c_out):
    # reference :30-34
    return nn.Sequential(nn.ConvTranspose2d(c_in, c_out, kernel_size=3, stride=2, padding=1, output_padding=1), nn.ReLU(inplace=True))


class DispNetS(nn.Module):
    def __init__(self, datasets='kitti'):
        super(DispNetS, self).__init__()
        if datasets == 'kitti':
            self.alpha, self.beta = 10, 0.01
        elif datasets == 'nyu':
            self.alpha, self.beta = 10, 0.1
        else:
            raise ValueError("undefined datasets %r" % (datasets,))
        cp = [32, 64, 128, 256, 512, 512, 512]
        self.conv1 = _downsample_conv(3, cp[0], 7)
        self.conv2 = _downsample_conv(cp[0], cp[1], 5)
        self.conv3 = _downsample_conv(cp[1], cp[2])
        self.conv4 = _downsample_conv(cp[2], cp[3])
        self.conv5 = _downsample_conv(cp[3], cp[4])
        self.conv6 = _downsample_conv(cp[4], cp[5])
        self.conv7 = _downsample_conv(cp[5], cp[6])
        up = [512, 512, 256, 128, 64, 32, 16]
        self.upconv7 = _upconv(cp[6], up[0])
        self.upconv6 = _upconv(up[0], up[1])
        self.upconv5 = _upconv(up[1], up[2])
        self.upconv4 = _upconv(up[2], up[3])
        self.upconv3 = _upconv(up[3], up[4])
        self.upconv2 = _upconv(up[4], up[5])
        self.upconv1 = _upconv(up[5], up[6])
        self.iconv7 = _conv(up[0] + cp[5], up[0])
        self.iconv6 = _conv(up[1] + cp[4], up[1])
        self.iconv5 = _conv(up[2] + cp[3], up[2])
        self.iconv4 = _conv(up[3] + cp[2], up[3])
        self.iconv3 = _conv(1 + up[4] + cp[1], up[4])
        self.iconv2 = _conv(1 + up[5] + cp[0], up[5])
        self.iconv1 = _conv(1 + up[6], up[6])
        self.predict_disp4 = _predict_disp(up[3])
        self.predict_disp3 = _predict_disp(up[4])
        self.predict_disp2 = _predict_disp(up[5])
        self.predict_disp1 = _predict_disp(up[6])
        self._rt = None

    def init_weights(self, use_pretrained_weights=False):
        # reference :86-91 (Conv2d / ConvTranspose2d only -- there are no other parametrised modules)
        xavier_init_like_reference(self)

    def forward(self, x):
        outs = run_net(self, x)
        return outs if self.training else outs[0]

    def _hot_parameters(self):
        return self.parameters()

    def _runtime(self):
        if self._rt is None:
            rt = {}
            for i in range(1, 8):
                seq = getattr(self, "conv%d" % i)
                rt["conv%d.0" % i] = engine.ConvLayer(seq[0])
                rt["conv%d.2" % i] = engine.ConvLayer(seq[2])
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
        relu = lambda name, pieces, out_hw=None: engine.block_conv_act(tape, sink, pieces, rt[name], ACT_RELU, out_hw=out_hw)
        head = lambda name, act: engine.block_conv_act(tape, sink, [P(act)], rt[name], ACT_SIGMOID_AFFINE, a, b)
        img = x
        enc, cur = [], img
        for i in range(1, 8):
            cur = relu("conv%d.0" % i, [P(cur)])
            cur = relu("conv%d.2" % i, [P(cur)])
            enc.append(cur)
        c1, c2, c3, c4, c5, c6, c7 = enc
        hw = lambda act: (act.H, act.W)
        i7 = relu("iconv7", [P(relu("upconv7", [P(c7)], hw(c6))), P(c6)])        # crop_like(upconv, skip), reference :101-103
        i6 = relu("iconv6", [P(relu("upconv6", [P(i7)], hw(c5))), P(c5)])
        i5 = relu("iconv5", [P(relu("upconv5", [P(i6)], hw(c4))), P(c4)])
        i4 = relu("iconv4", [P(relu("upconv4", [P(i5)], hw(c3))), P(c3)])
        d4 = head("predict_disp4", i4)
        i3 = relu("iconv3", [P(relu("upconv3", [P(i4)], hw(c2))), P(c2), P(engine.block_bilinear_up2(tape, d4, hw(c2)))])
        d3 = head("predict_disp3", i3)
        i2 = relu("iconv2", [P(relu("upconv2", [P(i3)], hw(c1))), P(c1), P(engine.block_bilinear_up2(tape, d3, hw(c1)))])
        d2 = head("predict_disp2", i2)
        i1 = relu("iconv1", [P(relu("upconv1", [P(i2)], hw(img))), P(engine.block_bilinear_up2(tape, d2, hw(img)))])
        d1 = head("predict_disp1", i1)
        return [d1, d2, d3, d4]
