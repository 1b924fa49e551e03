"""Mirror of the reference's networks/depth_decoder.py (monodepth2 DepthDecoder) on the HIP engine: per level
ReflectionPad+conv3x3+ELU, nearest x2, skip concat, ReflectionPad+conv3x3+ELU; disp = 0.01 + 9.99*sigmoid(conv3x3) --
reference :17-70, layers.py:106-136,193-196.  Reflection padding is a loader index map, upsample and concat are virtual, ELU
and the sigmoid affine live in the conv epilogue: 14 launches forward for the 10 ConvBlocks + 4 disparity heads."""
from collections import OrderedDict

import numpy as np
import torch.nn as nn

from .. import engine
from .._lib import ACT_ELU, ACT_SIGMOID_AFFINE
from ..layers import Conv3x3, ConvBlock
from ..models._common import run_net


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super(DepthDecoder, self).__init__()
        self.num_output_channels = num_output_channels
        self.use_skips = use_skips
        self.upsample_mode = 'nearest'
        self.scales = scales
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        self.convs = OrderedDict()
        for i in range(4, -1, -1):
            num_ch_in = self.num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            self.convs[("upconv", i, 0)] = ConvBlock(num_ch_in, self.num_ch_dec[i])
            num_ch_in = self.num_ch_dec[i]
            if self.use_skips and i > 0:
                num_ch_in += self.num_ch_enc[i - 1]
            self.convs[("upconv", i, 1)] = ConvBlock(num_ch_in, self.num_ch_dec[i])
        for s in self.scales:
            self.convs[("dispconv", s)] = Conv3x3(self.num_ch_dec[s], self.num_output_channels)
        self.decoder = nn.ModuleList(list(self.convs.values()))
        self.sigmoid = nn.Sigmoid()
        self._rt = None
        if num_output_channels != 1:
            raise NotImplementedError("the reference only ever builds DepthDecoder with num_output_channels=1")

    def forward(self, input_features):
        outs = run_net(self, *input_features)
        self.outputs = {("disp", s): o for s, o in zip(sorted(self.scales), outs)}
        if self.training:
            return self.outputs[("disp", 0)], self.outputs[("disp", 1)], self.outputs[("disp", 2)], self.outputs[("disp", 3)]
        return self.outputs[("disp", 0)]

    def _hot_parameters(self):
        return list(self.parameters())

    def _runtime(self):
        if self._rt is None:
            self._rt = {}
            for key, m in self.convs.items():
                conv = m.conv.conv if isinstance(m, ConvBlock) else m.conv
                self._rt[key] = engine.ConvLayer(conv, reflect_pad=1)
        return self._rt

    def _hip_decode(self, tape, sink, feats):
        rt = self._runtime()
        P = engine.Piece
        x = feats[-1]
        disps = {}
        for i in range(4, -1, -1):
            x = engine.block_conv_act(tape, sink, [P(x)], rt[("upconv", i, 0)], ACT_ELU)
            pieces = [P(x, up=True)]
            if self.use_skips and i > 0:
                pieces.append(P(feats[i - 1]))
            x = engine.block_conv_act(tape, sink, pieces, rt[("upconv", i, 1)], ACT_ELU)
            if i in self.scales:
                disps[i] = engine.block_conv_act(tape, sink, [P(x)], rt[("dispconv", i)], ACT_SIGMOID_AFFINE, 9.99, 0.01)
        return [disps[s] for s in sorted(self.scales)]

    def _hip_forward(self, tape, sink, *feats):
        return self._hip_decode(tape, sink, list(feats))
