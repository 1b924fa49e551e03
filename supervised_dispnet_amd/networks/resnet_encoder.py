"""Mirror of the reference's networks/resnet_encoder.py: (x - 0.45)/0.225, torchvision-layout ResNet-{18,34,50,101,152} trunk,
five feature maps (relu(bn1(conv1)), layer1(maxpool), layer2, layer3, layer4) -- reference :60-98."""
import numpy as np
import torch.nn as nn

from .. import engine
from ..models.Disp_res_50 import BasicBlock, Bottleneck, conv1x1, run_residual_block
from ..models._common import run_net

_CFG = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]), 50: (Bottleneck, [3, 4, 6, 3]),
        101: (Bottleneck, [3, 4, 23, 3]), 152: (Bottleneck, [3, 8, 36, 3])}
_URLS = {18: 'https://download.pytorch.org/models/resnet18-5c106cde.pth', 34: 'https://download.pytorch.org/models/resnet34-333f7ec4.pth',
         50: 'https://download.pytorch.org/models/resnet50-19c8e357.pth', 101: 'https://download.pytorch.org/models/resnet101-5d3b4d8f.pth',
         152: 'https://download.pytorch.org/models/resnet152-b121ed2d.pth'}


class ResNetContainer(nn.Module):
    """Parameter container laid out like torchvision.models.ResNet (conv1, bn1, relu, maxpool, layer1..4, avgpool, fc)."""

    def __init__(self, block, layers, num_input_images=1):
        super(ResNetContainer, self).__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(num_input_images * 3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, 1000)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride), nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)


class ResnetEncoder(nn.Module):
    def __init__(self, num_layers, pretrained, num_input_images=1):
        super(ResnetEncoder, self).__init__()
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        if num_layers not in _CFG:
            raise ValueError("{} is not a valid number of resnet layers".format(num_layers))
        if num_input_images > 1 and num_layers not in (18, 50):
            raise AssertionError("Can only run with 18 or 50 layer resnet")
        block, layers = _CFG[num_layers]
        self.encoder = ResNetContainer(block, layers, num_input_images)
        if pretrained:
            import torch
            import torch.utils.model_zoo as model_zoo
            loaded = model_zoo.load_url(_URLS[num_layers])
            if num_input_images > 1:
                loaded['conv1.weight'] = torch.cat([loaded['conv1.weight']] * num_input_images, 1) / num_input_images
            self.encoder.load_state_dict(loaded)
        if num_layers > 34:
            self.num_ch_enc[1:] *= 4
        self._rt = None

    def forward(self, input_image):
        self.features = list(run_net(self, input_image))
        return self.features

    def _hot_parameters(self):
        return [p for n, p in self.named_parameters() if ".fc." not in n]

    def _hip_features(self, tape, sink, x):
        enc = self.encoder
        if self._rt is None:
            self._rt = engine.ConvLayer(enc.conv1)
        xn = engine.normalize_input(x.t, 0.45, 0.225)
        y = engine.block_conv_bn(tape, sink, engine.Piece(engine.Act.from_nchw_image(xn)), self._rt, enc.bn1, self.training)
        f0 = engine.block_bn_relu(tape, y)
        feats = [f0]
        cur = engine.block_maxpool3s2(tape, f0)
        for layer in (enc.layer1, enc.layer2, enc.layer3, enc.layer4):
            for blk in layer:
                cur = run_residual_block(tape, sink, cur, blk, self.training)
            feats.append(cur)
        return feats

    def _hip_forward(self, tape, sink, x):
        return self._hip_features(tape, sink, x)
