"""Drop-in mirror of the reference's models/Disp_vgg_feature.py -- what `--network disp_vgg` builds (train.py:248): the Disp_vgg
decoder on a torchvision vgg16 held whole as `self.features` (state_dict keys `features.features.{0,2,5,...,28}.*` plus the never-used
`features.classifier.*`), sliced [0:5], [5:10], [10:17], [17:24], [24:31] in forward (reference :138-142).  `with_classifier=False`
drops the 123.6 M unused classifier parameters (they never take part in compute, the arena, the all-reduce or Adam either way)."""
import torch.nn as nn

from ._common import VGG16Container, xavier_init_like_reference
from .Disp_vgg import _VggDispBase

_SLICES = ((0, 5), (5, 10), (10, 17), (17, 24), (24, 31))


class Disp_vgg_feature(_VggDispBase):
    def __init__(self, datasets='kitti', use_pretrained_weights=False, with_classifier=True):
        super(Disp_vgg_feature, self).__init__()
        self.use_pretrained_weights = use_pretrained_weights
        self.only_train_dec = False
        if datasets == 'kitti':
            self.alpha, self.beta = 10, 0.01
        elif datasets == 'nyu':
            self.alpha, self.beta = 10, 0.1
        else:
            raise ValueError("undefined datasets %r" % (datasets,))
        self.features = VGG16Container(with_classifier)
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
        own = self.features.state_dict()
        own.update({k: v for k, v in params.items() if k in own})
        self.features.load_state_dict(own)

    def _hot_parameters(self):
        return [p for name, p in self.named_parameters() if ".classifier." not in name]

    def _encoder_convs(self):
        f = self.features.features
        return [[f[i] for i in range(lo, hi) if isinstance(f[i], nn.Conv2d)] for lo, hi in _SLICES]
