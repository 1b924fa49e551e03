"""Drop-in mirror of the reference's models/Disp_vgg_BN_DORN.py: the Disp_vgg_BN trunk (without disp0) + Dropout2d(0.5) +
1x1 conv to 2*ordinal_c logits + OrdinalRegressionLayer.  Returns (decode_c int64 [N,1,H,W], ord_c1 f32 [N,K,H,W]) in both
training and eval mode -- reference models/Disp_vgg_BN_DORN.py:72-227.
"""
import torch
import torch.nn as nn

from .. import _lib, engine
from .._lib import ACT_NONE
from .Disp_vgg_BN import Disp_vgg_BN, _iconv, _predict_disp, _upconv  # noqa: F401
from ._common import run_net


class _Ordinal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        engine.require_cuda(x, "ordinal logits")
        if x.dtype != torch.float32:
            raise TypeError("expected float32 logits")
        n, c2, h, w = x.shape
        if c2 % 2:
            raise ValueError("OrdinalRegressionLayer needs an even number of channels, got %d" % c2)
        k = c2 // 2
        sn, sc, sh, sw = x.stride()
        if sh != w * sw:                                   # pixels must be addressable with one stride
            x = x.contiguous()
            sn, sc, sh, sw = x.stride()
        ordc = torch.empty((n, k, h, w), dtype=torch.float32, device=x.device)
        dec = torch.empty((n, 1, h, w), dtype=torch.int64, device=x.device)
        _lib.call("dn_ordinal_fwd", x.data_ptr(), sn, sw, sc, n, h * w, k, ordc.data_ptr(), dec.data_ptr(), engine._stream())
        ctx.save_for_backward(x, ordc)
        ctx.mark_non_differentiable(dec)
        return dec, ordc

    @staticmethod
    def backward(ctx, _ddec, dord):
        x, ordc = ctx.saved_tensors
        n, c2, h, w = x.shape
        sn, sc, sh, sw = x.stride()
        g = dord.contiguous().float()
        dx = torch.empty_strided(x.shape, x.stride(), dtype=torch.float32, device=x.device)
        _lib.call("dn_ordinal_bwd", x.data_ptr(), sn, sw, sc, ordc.data_ptr(), g.data_ptr(), n, h * w, c2 // 2, dx.data_ptr(), engine._stream())
        return dx


class OrdinalRegressionLayer(nn.Module):
    """reference models/Disp_vgg_BN_DORN.py:196-227: x [N,2K,H,W] -> (decode_c [N,1,H,W] int64, ord_c1 [N,K,H,W]).
    The logits are clamped to [1e-8, 1e8] before the 2-way softmax exactly like the reference (which makes negative logits
    behave like 1e-8 and blocks their gradient)."""

    def forward(self, x):
        return _Ordinal.apply(x)


class Disp_vgg_BN_DORN(Disp_vgg_BN):
    def __init__(self, datasets='kitti', ordinal_c=71, with_classifier=True):
        super(Disp_vgg_BN_DORN, self).__init__(datasets, with_classifier)
        del self.disp0                                      # the DORN variant has no full-resolution disparity head (:108-110)
        self.dropout = nn.Dropout2d(p=0.5)
        self.conv_ord = nn.Conv2d(16, 2 * ordinal_c, 1)
        self.orl = OrdinalRegressionLayer()
        self.fused_head = True                              # False: conv -> logits in HBM -> OrdinalRegressionLayer (the r01 path)
        self._dropout_mask = None                           # test hook: fixed [N,16] keep/scale mask instead of RNG

    def forward(self, x):
        outs = run_net(self, x)
        if len(outs) == 2:                                  # fused head: (ord_c1, decode_c) straight from the engine
            return outs[1], outs[0]
        return self.orl(outs[0])                            # [N,2K,H,W] view of the NHWC logits -> OrdinalRegressionLayer

    def _runtime(self):
        rt = super(Disp_vgg_BN_DORN, self)._runtime()
        if "conv_ord" not in rt:
            rt["conv_ord"] = engine.ConvLayer(self.conv_ord)
        return rt

    def _grad_production_order(self):
        return [self.conv_ord.bias, self.conv_ord.weight] + super(Disp_vgg_BN_DORN, self)._grad_production_order()

    def _hip_forward(self, tape, sink, x):
        feats = self._encoder(tape, sink, x)
        i0, _d1, _d2, _d3, _head = self._decoder_trunk(tape, sink, feats)
        cur, mask = i0, None
        if self.training and self.dropout.p > 0:
            mask = self._dropout_mask
            if mask is None:
                keep = 1.0 - self.dropout.p
                mask = torch.bernoulli(torch.full((i0.N, i0.C), keep, dtype=torch.float32, device=i0.t.device)) / keep
            mask = mask.to(i0.t.device).contiguous().float()
        if self.fused_head and engine.ord_head_fusable(i0, self.conv_ord.out_channels // 2):
            o, d = engine.block_ord_head(tape, sink, i0, self.conv_ord, mask)
            return [o, d]
        if mask is not None:
            cur = engine.block_channel_scale(tape, i0, mask)
        pre = engine.block_conv_act(tape, sink, [engine.Piece(cur)], self._runtime()["conv_ord"], ACT_NONE)
        return [pre]
