"""Drop-in mirror of the reference's models/Disp_res_18.py: the Disp_res_50 encoder-decoder with torchvision's ResNet-18
BasicBlocks (two 3x3 convolutions, expansion 1) -- reference models/Disp_res_18.py:50-135 (network), :249-286 (BasicBlock).

Same state_dict keys as the reference (conv1, bn1, layer{1..4}.{0,1}.{conv,bn}{1,2}, layer{2..4}.0.downsample.{0,1}, upconv{5..1}.0,
iconv{5..1}.0, predict_disp{4..1}.0), same forward contract, same bn1 quirk (:141-145 there: bn1 evaluated, output discarded).
The schedule is Disp_res_50's (`_hip_forward`): only the block type, the per-stage block counts and the channel expansion differ.
"""
from .Disp_res_50 import BasicBlock, Disp_res_50


class Disp_res_18(Disp_res_50):
    _block, _expansion, _counts = BasicBlock, 1, (2, 2, 2, 2)
    _pretrained_url = 'https://download.pytorch.org/models/resnet18-5c106cde.pth'
