"""Drop-in mirror of the reference's models/Disp_vgg_BN.py (the north-star network) on hand-written HIP kernels.

Same constructor signature, attributes (`alpha`, `beta`, `only_train_dec`), `init_weights(use_pretrained_weights)`,
state_dict keys / shapes (125 keys incl. the unused VGG classifier), forward contract: tuple (disp0..disp3) in
training mode, disp0 alone in eval mode -- reference models/Disp_vgg_BN.py:72-191.
"""
import torch.nn as nn

from .. import engine
from .._lib import ACT_LEAKY, ACT_SIGMOID_AFFINE
from ._common import VGG16BNContainer, VGG_STAGES, run_net, xavier_init_like_reference


def _upconv(c_in, c_out):
    # ConvTranspose2dBlock1(c_in, c_out, 4, 2, 1, 0): ConvTranspose2d + LeakyReLU(0.1)   (reference :53-64)
    return nn.Sequential(nn.ConvTranspose2d(c_in, c_out, 4, 2, 1, 0), nn.LeakyReLU(0.1))


def _iconv(c_in, c_out):
    # Conv2dBlock1(c_in, c_out, 3, 1, 1): Conv2d + LeakyReLU(0.1)   (reference :40-50)
    return nn.Sequential(nn.Conv2d(c_in, c_out, 3, 1, 1), nn.LeakyReLU(0.1))


def _predict_disp(c_in):
    # reference :66-70
    return nn.Sequential(nn.Conv2d(c_in, 1, kernel_size=3, padding=1), nn.Sigmoid())


class Disp_vgg_BN(nn.Module):
    def __init__(self, datasets='kitti', with_classifier=True):
        super(Disp_vgg_BN, self).__init__()
        self.only_train_dec = False
        if datasets == 'kitti':
            self.alpha, self.beta = 10, 0.01
        elif datasets == 'nyu':
            self.alpha, self.beta = 10, 0.1
        else:
            raise ValueError("undefined datasets %r" % (datasets,))
        self.features = VGG16BNContainer(with_classifier)
        self.upconv4, self.iconv4 = _upconv(512, 256), _iconv(256 + 512, 256)
        self.upconv3, self.iconv3 = _upconv(256, 128), _iconv(128 + 256, 128)
        self.upconv2, self.iconv2 = _upconv(128, 64), _iconv(64 + 128 + 1, 64)
        self.upconv1, self.iconv1 = _upconv(64, 32), _iconv(32 + 64 + 1, 32)
        self.upconv0, self.iconv0 = _upconv(32, 16), _iconv(16 + 1, 16)
        self.disp3, self.disp2, self.disp1, self.disp0 = _predict_disp(128), _predict_disp(64), _predict_disp(32), _predict_disp(16)
        self._rt = None

    # ------------------------------------------------------------------ reference API
    def init_weights(self, use_pretrained_weights=False):
        xavier_init_like_reference(self)
        if use_pretrained_weights:
            import torch.utils.model_zoo as model_zoo
            print("loading pretrained weights downloaded from pytorch.org")
            self.load_vgg_params(model_zoo.load_url('https://download.pytorch.org/models/vgg16_bn-6c64b313.pth'))
        else:
            print("do not load pretrained weights for the monocular model")

    def load_vgg_params(self, params):
        own = self.features.state_dict()
        own.update({k: v for k, v in params.items() if k in own})
        self.features.load_state_dict(own)

    def forward(self, x):
        outs = run_net(self, x)
        return outs if self.training else outs[0]

    # ------------------------------------------------------------------ engine side
    def _hot_parameters(self):
        """Parameters that take part in forward/backward (everything except features.classifier.*)."""
        for name, p in self.named_parameters():
            if ".classifier." not in name:
                yield p

    def _grad_production_order(self):
        """Parameters in the order backward writes their gradients (decoder first, encoder stage 5 -> 1): the arena and
        the all-reduce buckets follow it so each bucket can leave as soon as it is complete."""
        order = []
        for name in ("disp0", "iconv0", "upconv0", "disp1", "iconv1", "upconv1", "disp2", "iconv2", "upconv2", "disp3",
                     "iconv3", "upconv3", "iconv4", "upconv4"):
            if hasattr(self, name):
                m = getattr(self, name)[0]
                order += [m.bias, m.weight]
        f = self.features.features
        for i in reversed(range(len(f))):
            if isinstance(f[i], nn.Conv2d):
                order += [f[i + 1].weight, f[i + 1].bias, f[i].bias, f[i].weight]
        return order

    def _runtime(self):
        if self._rt is None:
            f = self.features.features
            rt = {"enc": []}
            for lo, hi in VGG_STAGES:
                stage = [(engine.ConvLayer(f[i]), f[i + 1]) for i in range(lo, hi) if isinstance(f[i], nn.Conv2d)]
                rt["enc"].append(stage)
            for name in ("upconv4", "upconv3", "upconv2", "upconv1", "upconv0"):
                rt[name] = engine.ConvLayer(getattr(self, name)[0], transposed=True)
            for name in ("iconv4", "iconv3", "iconv2", "iconv1", "iconv0", "disp3", "disp2", "disp1", "disp0"):
                if hasattr(self, name):
                    rt[name] = engine.ConvLayer(getattr(self, name)[0])
            self._rt = rt
        return self._rt

    def _encoder(self, tape, sink, x):
        rt = self._runtime()
        cur = engine.Piece(x)
        feats = []
        for stage in rt["enc"]:
            for layer, bn in stage:
                cur = engine.Piece(engine.block_conv_bn(tape, sink, cur, layer, bn, self.training))
            pooled = engine.block_pool(tape, cur.act)
            if self.only_train_dec:
                pooled.needs_grad = False            # .detach() of the five stage outputs (reference :149-154)
            feats.append(pooled)
            cur = engine.Piece(pooled)
        return feats

    def _decoder_trunk(self, tape, sink, feats):
        rt = self._runtime()
        P = engine.Piece
        c1, c2, c3, c4, c5 = feats
        a, b = float(self.alpha), float(self.beta)
        lrelu = lambda name, pieces: engine.block_conv_act(tape, sink, pieces, rt[name], ACT_LEAKY, 0.1)
        head = lambda name, act: engine.block_conv_act(tape, sink, [P(act)], rt[name], ACT_SIGMOID_AFFINE, a, b)
        up4 = lrelu("upconv4", [P(c5)])
        i4 = lrelu("iconv4", [P(up4), P(c4)])
        up3 = lrelu("upconv3", [P(i4)])
        i3 = lrelu("iconv3", [P(up3), P(c3)])
        d3 = head("disp3", i3)
        up2 = lrelu("upconv2", [P(i3)])
        i2 = lrelu("iconv2", [P(up2), P(c2), P(d3, up=True)])
        d2 = head("disp2", i2)
        up1 = lrelu("upconv1", [P(i2)])
        i1 = lrelu("iconv1", [P(up1), P(c1), P(d2, up=True)])
        d1 = head("disp1", i1)
        up0 = lrelu("upconv0", [P(i1)])
        i0 = lrelu("iconv0", [P(up0), P(d1, up=True)])
        return i0, d1, d2, d3, head

    def _hip_forward(self, tape, sink, x):
        feats = self._encoder(tape, sink, x)
        i0, d1, d2, d3, head = self._decoder_trunk(tape, sink, feats)
        d0 = head("disp0", i0)
        return [d0, d1, d2, d3]
