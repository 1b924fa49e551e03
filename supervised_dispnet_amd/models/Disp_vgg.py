"""Drop-in mirrors of the reference's BatchNorm-free VGG16 DispNets: models/Disp_vgg.py (own conv1..conv5 Sequentials) and
models/Disp_vgg_feature.py (a whole torchvision vgg16 held as `self.features`, the net `--network disp_vgg` builds, train.py:248).

Encoder: five stages of conv3x3 + ReLU (2, 2, 3, 3, 3 layers), each ENDING in MaxPool2d(2, 2); the skips are the pooled stage
outputs (reference Disp_vgg.py:79-100,157-173).  Decoder: ConvTranspose2d(4, 2, 1) + ReLU, concat, conv3x3 + ReLU (leaky=False
is the default of Conv2dBlock1 / ConvTranspose2dBlock1 there, :36-59), disparity heads alpha*sigmoid+beta, and the function NAMED
upsample_nn_nearest is `F.interpolate(..., mode='bilinear', align_corners=False)` (:8-9) -- bilinear it is.
`only_train_dec` detaches the encoder only when `use_pretrained_weights` is also set (:163).
"""
import torch.nn as nn

from .. import engine
from .._lib import ACT_RELU, ACT_SIGMOID_AFFINE
from ._common import run_net, xavier_init_like_reference

_STAGE_CFG = ((64, 64), (128, 128), (256, 256, 256), (512, 512, 512), (512, 512, 512))


def _upconv(c_in, c_out):
    return nn.Sequential(nn.ConvTranspose2d(c_in, c_out, 4, 2, 1, 0), nn.ReLU(inplace=True))


def _iconv(c_in, c_out):
    return nn.Sequential(nn.Conv2d(c_in, c_out, 3, 1, 1), nn.ReLU(inplace=True))


def _predict_disp(c_in):
    return nn.Sequential(nn.Conv2d(c_in, 1, kernel_size=3, padding=1), nn.Sigmoid())


def _encoder_stage(cfg, c_in):
    layers = []
    for v in cfg:
        layers += [nn.Conv2d(c_in, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
        c_in = v
    layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
    return nn.Sequential(*layers)


class _VggDispBase(nn.Module):
    """Decoder + schedule shared by the two mirrors; subclasses provide `_encoder_convs()` -> five lists of nn.Conv2d."""

    def _build_decoder(self):
        self.upconv4, self.iconv4 = _upconv(512, 256), _iconv(256 + 512, 256)
        self.upconv3, self.iconv3 = _upconv(256, 128), _iconv(128 + 256, 128)
        self.upconv2, self.iconv2 = _upconv(128, 64), _iconv(64 + 128 + 1, 64)
        self.upconv1, self.iconv1 = _upconv(64, 32), _iconv(32 + 64 + 1, 32)
        self.upconv0, self.iconv0 = _upconv(32, 16), _iconv(16 + 1, 16)
        self.disp3, self.disp2, self.disp1, self.disp0 = _predict_disp(128), _predict_disp(64), _predict_disp(32), _predict_disp(16)
        self._rt = None

    def forward(self, x):
        outs = run_net(self, x)
        return outs if self.training else outs[0]

    def _grad_production_order(self):
        order = []
        for name in ("disp0", "iconv0", "upconv0", "disp1", "iconv1", "upconv1", "disp2", "iconv2", "upconv2", "disp3",
                     "iconv3", "upconv3", "iconv4", "upconv4"):
            m = getattr(self, name)[0]
            order += [m.bias, m.weight]
        for stage in reversed(self._encoder_convs()):
            for m in reversed(stage):
                order += [m.bias, m.weight]
        return order

    def _runtime(self):
        if self._rt is None:
            rt = {"enc": [[engine.ConvLayer(m) for m in stage] for stage in self._encoder_convs()]}
            for name in ("upconv4", "upconv3", "upconv2", "upconv1", "upconv0"):
                rt[name] = engine.ConvLayer(getattr(self, name)[0], transposed=True)
            for name in ("iconv4", "iconv3", "iconv2", "iconv1", "iconv0", "disp3", "disp2", "disp1", "disp0"):
                rt[name] = engine.ConvLayer(getattr(self, name)[0])
            self._rt = rt
        return self._rt

    def _hip_forward(self, tape, sink, x):
        rt = self._runtime()
        P = engine.Piece
        a, b = float(self.alpha), float(self.beta)
        relu = lambda layer, pieces: engine.block_conv_act(tape, sink, pieces, layer, ACT_RELU)
        head = lambda name, act: engine.block_conv_act(tape, sink, [P(act)], rt[name], ACT_SIGMOID_AFFINE, a, b)
        bil = lambda d: engine.block_bilinear_up2(tape, d, (2 * d.H, 2 * d.W))
        feats, cur = [], x
        for stage in rt["enc"]:
            for layer in stage:
                cur = relu(layer, [P(cur)])
            cur = engine.block_maxpool2(tape, cur)
            feats.append(cur)
        if self.use_pretrained_weights and self.only_train_dec:       # reference models/Disp_vgg.py:163-168
            for f in feats:
                f.needs_grad = False
        c1, c2, c3, c4, c5 = feats
        i4 = relu(rt["iconv4"], [P(relu(rt["upconv4"], [P(c5)])), P(c4)])
        i3 = relu(rt["iconv3"], [P(relu(rt["upconv3"], [P(i4)])), P(c3)])
        d3 = head("disp3", i3)
        i2 = relu(rt["iconv2"], [P(relu(rt["upconv2"], [P(i3)])), P(c2), P(bil(d3))])
        d2 = head("disp2", i2)
        i1 = relu(rt["iconv1"], [P(relu(rt["upconv1"], [P(i2)])), P(c1), P(bil(d2))])
        d1 = head("disp1", i1)
        i0 = relu(rt["iconv0"], [P(relu(rt["upconv0"], [P(i1)])), P(bil(d1))])
        d0 = head("disp0", i0)
        return [d0, d1, d2, d3]


class Disp_vgg(_VggDispBase):
    """reference models/Disp_vgg.py:71-207: `Disp_vgg(alpha=10, beta=0.01, use_pretrained_weights=False)`; state_dict keys
    conv{1..5}.{0,2[,4]}.*, upconv{4..0}.0.*, iconv{4..0}.0.*, disp{3..0}.0.*."""

    def __init__(self, alpha=10, beta=0.01, use_pretrained_weights=False):
        super(Disp_vgg, self).__init__()
        self.use_pretrained_weights = use_pretrained_weights
        self.only_train_dec = False
        self.alpha, self.beta = alpha, beta
        c = 3
        for i, cfg in enumerate(_STAGE_CFG, start=1):
            setattr(self, "conv%d" % i, _encoder_stage(cfg, c))
            c = cfg[-1]
        self._build_decoder()

    def init_weights(self, use_pretrained_weights=False):
        xavier_init_like_reference(self)
        if use_pretrained_weights:
            import torch.utils.model_zoo as model_zoo
            print("loading pretrained weights downloaded from pytorch.org")
            self.load_vgg_params(model_zoo.load_url('https://download.pytorch.org/models/vgg16-397923af.pth'))
        else:
            print("do not load pretrained weights for the monocular model")

    def load_vgg_params(self, params):
        """torchvision vgg16 `features.N` -> conv{1..5}.{0,2,4} (reference :137-154)."""
        transfer = {"conv1": {0: 0, 2: 2}, "conv2": {0: 5, 2: 7}, "conv3": {0: 10, 2: 12, 4: 14}, "conv4": {0: 17, 2: 19, 4: 21},
                    "conv5": {0: 24, 2: 26, 4: 28}}
        for name, cfg in transfer.items():
            sd = {}
            for to_id, from_id in cfg.items():
                sd["%d.weight" % to_id] = params["features.%d.weight" % from_id]
                sd["%d.bias" % to_id] = params["features.%d.bias" % from_id]
            getattr(self, name).load_state_dict(sd)

    def _hot_parameters(self):
        return list(self.parameters())

    def _encoder_convs(self):
        return [[m for m in getattr(self, "conv%d" % i) if isinstance(m, nn.Conv2d)] for i in range(1, 6)]
