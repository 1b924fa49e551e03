"""Drop-in mirrors of the reference's models/ASPP.py (`deeplab_depth`: ResNet-101 with dilated layer3 / layer4 + ASPP classifier) and
models/res_aspp.py (`res50_aspp`: the same file with a ResNet-50 trunk) on the HIP engine.

Same state_dict keys (`Scale.conv1`, `Scale.bn1`, `Scale.layer{1..4}.*`, `Scale.layer5.conv2d_list.{0..3}`), `init_weights`, forward
contract ([map at the input size] in training mode, the tensor in eval mode).  Reference facts kept (models/ASPP.py):
  * the bottleneck's STRIDE sits on its first 1x1 convolution (:59), its 3x3 is dilated (layer3: 2, layer4: 4, :62-72);
  * every layer's first block has a downsample branch when dilated (:147-152);
  * the max-pool uses ceil_mode=True (:138);
  * every BatchNorm's affine pair is frozen (requires_grad False, :61-63,74-82,133-135,153-155) but the layers still normalise with
    batch statistics in training mode;
  * the classifier sums four dilated 3x3 convolutions 2048 -> 1 (dilations 6, 12, 18, 24) before 10 * sigmoid + 0.01 (:107-123);
  * `deeplab_depth.__init__` takes no `datasets` argument (the reference's train.py:255 passes one and fails); this mirror accepts
    and ignores it so that `--network ASPP` runs.
"""
import torch.nn as nn

from .. import engine
from .._lib import ACT_SIGMOID_AFFINE
from ._common import run_net
from .Disp_res_50 import run_residual_block

affine_par = True


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation_=1, downsample=None):
        super(Bottleneck, self).__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, stride=stride, bias=False)
        self.bn1 = nn.BatchNorm2d(planes, affine=affine_par)
        padding = {1: 1, 2: 2, 4: 4}[dilation_]
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=padding, bias=False, dilation=dilation_)
        self.bn2 = nn.BatchNorm2d(planes, affine=affine_par)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4, affine=affine_par)
        for bn in (self.bn1, self.bn2, self.bn3):
            for q in bn.parameters():
                q.requires_grad = False
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


class Classifier_Module(nn.Module):
    def __init__(self, dilation_series, padding_series):
        super(Classifier_Module, self).__init__()
        self.conv2d_list = nn.ModuleList()
        for dilation, padding in zip(dilation_series, padding_series):
            self.conv2d_list.append(nn.Conv2d(2048, 1, kernel_size=3, stride=1, padding=padding, dilation=dilation, bias=True))
        for m in self.conv2d_list:
            m.weight.data.normal_(0, 0.01)
        self.Sigmoid = nn.Sigmoid()


class ResNet(nn.Module):
    def __init__(self, block, layers):
        self.inplanes = 64
        super(ResNet, self).__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64, affine=affine_par)
        for q in self.bn1.parameters():
            q.requires_grad = False
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1, ceil_mode=True)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=1, dilation__=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=1, dilation__=4)
        self.layer5 = Classifier_Module([6, 12, 18, 24], [6, 12, 18, 24])

    def _make_layer(self, block, planes, blocks, stride=1, dilation__=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion or dilation__ == 2 or dilation__ == 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion, affine=affine_par))
        for q in downsample._modules['1'].parameters():
            q.requires_grad = False
        layers = [block(self.inplanes, planes, stride, dilation_=dilation__, downsample=downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, dilation_=dilation__))
        return nn.Sequential(*layers)


class _ASPPDepth(nn.Module):
    _counts = (3, 4, 23, 3)
    _tv = "resnet101"

    def __init__(self, datasets='kitti'):
        super(_ASPPDepth, self).__init__()
        self.Scale = ResNet(Bottleneck, list(self._counts))
        self._rt = None

    def forward(self, x):
        outs = run_net(self, x)
        return [outs[0]] if self.training else outs[0]

    def init_weights(self, use_pretrained_weights=False):
        """reference :198-213: every Conv2d weight ~ N(0, 0.01), BatchNorm gamma 1 / beta 0."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, 0.01)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        if use_pretrained_weights:
            import torchvision.models as tvm
            print("loading pretrained weights downloaded from pytorch.org")
            sd = getattr(tvm, self._tv)(pretrained=True).state_dict()
            self.load_state_dict({'Scale.' + k: v for k, v in sd.items()}, strict=False)
        else:
            print("do not load pretrained weights for the monocular model")

    # ------------------------------------------------------------------ engine side
    def _hot_parameters(self):
        """The trainable ones: the frozen BatchNorm pairs take no gradient (their layers still compute one for their inputs)."""
        return [q for q in self.parameters() if q.requires_grad]

    def _grad_production_order(self):
        S = self.Scale
        order = []
        for m in S.layer5.conv2d_list:
            order += [m.bias, m.weight]
        for layer in (S.layer4, S.layer3, S.layer2, S.layer1):
            for blk in reversed(list(layer)):
                if blk.downsample is not None:
                    order.append(blk.downsample[0].weight)
                order += [blk.conv3.weight, blk.conv2.weight, blk.conv1.weight]
        order.append(S.conv1.weight)
        return order

    def _runtime(self):
        if self._rt is None:
            S = self.Scale
            self._rt = {"conv1": engine.ConvLayer(S.conv1), "aspp": [engine.ConvLayer(m) for m in S.layer5.conv2d_list]}
        return self._rt

    def _hip_forward(self, tape, sink, x):
        rt, S = self._runtime(), self.Scale
        training = self.training
        cur = engine.block_conv_bn(tape, sink, engine.Piece(x), rt["conv1"], S.bn1, training)
        cur = engine.block_maxpool3s2(tape, engine.block_bn_relu(tape, cur), ceil_mode=True)
        for layer in (S.layer1, S.layer2, S.layer3, S.layer4):
            for blk in layer:
                cur = run_residual_block(tape, sink, cur, blk, training)
        head = engine.block_sum_convs_act(tape, sink, cur, rt["aspp"], ACT_SIGMOID_AFFINE, 10.0, 0.01)
        return [engine.block_resize_bilinear(tape, head, (x.H, x.W), align_corners=True)]


class deeplab_depth(_ASPPDepth):
    """models/ASPP.py:184-228 (ResNet-101 trunk)."""
    _counts, _tv = (3, 4, 23, 3), "resnet101"


class res50_aspp(_ASPPDepth):
    """models/res_aspp.py:184-228 (ResNet-50 trunk)."""
    _counts, _tv = (3, 4, 6, 3), "resnet50"
