"""Mirror of the reference's networks/vgg_encoder.py: (x - 0.45)/0.225, then the five VGG16-BN stages; returns the five
pooled feature maps [64,128,256,512,512] channels (reference :60-87)."""
import numpy as np
import torch.nn as nn

from .. import engine
from ..models._common import VGG16BNContainer, VGG_STAGES, run_net


class vggEncoder(nn.Module):
    def __init__(self, num_layers, pretrained, num_input_images=1):
        super(vggEncoder, self).__init__()
        self.num_ch_enc = np.array([64, 128, 256, 512, 512])
        if num_layers != 16:
            raise ValueError("{} is not a valid number of vgg layers".format(num_layers))
        if num_input_images > 1:
            raise NotImplementedError("multi-image VGG input is unfinished in the reference too (networks/vgg_encoder.py:72-74)")
        self.encoder = VGG16BNContainer(with_classifier=True)
        if pretrained:
            import torch.utils.model_zoo as model_zoo
            self.encoder.load_state_dict(model_zoo.load_url('https://download.pytorch.org/models/vgg16_bn-6c64b313.pth'))
        self._rt = None

    def forward(self, input_image):
        self.features = list(run_net(self, input_image))
        return self.features

    def _hot_parameters(self):
        return [p for n, p in self.named_parameters() if ".classifier." not in n]

    def _runtime(self):
        if self._rt is None:
            f = self.encoder.features
            self._rt = [[(engine.ConvLayer(f[i]), f[i + 1]) for i in range(lo, hi) if isinstance(f[i], nn.Conv2d)] for lo, hi in VGG_STAGES]
        return self._rt

    def _hip_features(self, tape, sink, x):
        """x: Act over the user image -> list of 5 plain activations."""
        xn = engine.normalize_input(x.t, 0.45, 0.225)
        cur = engine.Piece(engine.Act.from_nchw_image(xn))
        feats = []
        for stage in self._runtime():
            for layer, bn in stage:
                cur = engine.Piece(engine.block_conv_bn(tape, sink, cur, layer, bn, self.training))
            pooled = engine.block_pool(tape, cur.act)
            feats.append(pooled)
            cur = engine.Piece(pooled)
        return feats

    def _hip_forward(self, tape, sink, x):
        return self._hip_features(tape, sink, x)
