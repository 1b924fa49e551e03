"""Drop-in mirror of the reference's models/Disp_res_101.py: the six-level Disp_res network with ResNet-101's 23-block layer3 and a
ReLU (not LeakyReLU) decoder -- reference models/Disp_res_101.py:22-33 (conv / upconv with ReLU), :66 (23 blocks), :129-196 (forward,
the same text as models/Disp_res.py:137-208).  Everything else, the H/4 crop quirk included, is documented in Disp_res.py here."""
from .Disp_res import Disp_res


class Disp_res_101(Disp_res):
    _layer3_blocks = 23
    _leaky = False
    _pretrained_url = 'https://download.pytorch.org/models/resnet101-5d3b4d8f.pth'
