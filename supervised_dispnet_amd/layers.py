"""Mirror of the hot-path pieces of the reference's layers.py on HIP kernels: SSIM, get_smooth_loss, disp_to_depth, and the
monodepth2 building blocks (ConvBlock / Conv3x3 / upsample) used by networks.DepthDecoder.
"""
import torch
import torch.nn as nn

from . import _lib
from .engine import _stream, require_cuda


def disp_to_depth(disp, min_depth, max_depth):
    """reference layers.py:14-24 (host-side scalar arithmetic on the network output; not a kernel in the reference either)."""
    min_disp = 1 / max_depth
    max_disp = 1 / min_depth
    scaled_disp = min_disp + (max_disp - min_disp) * disp
    depth = 1 / scaled_disp
    return scaled_disp, depth


class _SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        require_cuda(x, "SSIM input x")
        require_cuda(y, "SSIM input y")
        xc, yc = x.contiguous().float(), y.contiguous().float()
        if xc.shape != yc.shape or xc.dim() != 4:
            raise ValueError("SSIM expects two [B,C,H,W] tensors of the same shape")
        b, c, h, w = xc.shape
        out = torch.empty_like(xc)
        _lib.call("dn_ssim_fwd", xc.data_ptr(), yc.data_ptr(), b * c, h, w, out.data_ptr(), _stream())
        ctx.save_for_backward(xc, yc)
        return out

    @staticmethod
    def backward(ctx, dout):
        xc, yc = ctx.saved_tensors
        b, c, h, w = xc.shape
        g = dout.contiguous().float()
        ws = torch.empty(5 * xc.numel(), dtype=torch.float32, device=xc.device)
        dx = torch.empty_like(xc) if ctx.needs_input_grad[0] else None
        dy = torch.empty_like(yc) if ctx.needs_input_grad[1] else None
        _lib.call("dn_ssim_bwd", xc.data_ptr(), yc.data_ptr(), g.data_ptr(), b * c, h, w, ws.data_ptr(),
                  dx.data_ptr() if dx is not None else None, dy.data_ptr() if dy is not None else None, _stream())
        return dx, dy


class SSIM(nn.Module):
    """reference layers.py:215-245: ReflectionPad2d(1) + 3x3 means, C1 = 0.01^2, C2 = 0.03^2, clamp((1 - SSIM)/2, 0, 1)."""

    def __init__(self):
        super(SSIM, self).__init__()
        self.C1 = 0.01 ** 2
        self.C2 = 0.03 ** 2

    def forward(self, x, y):
        return _SSIM.apply(x, y)


class _EdgeSmooth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, img):
        require_cuda(disp, "disparity")
        require_cuda(img, "image")
        dc, ic = disp.contiguous().float(), img.contiguous().float()
        b, c1, h, w = dc.shape
        if c1 != 1:
            raise ValueError("get_smooth_loss expects a 1-channel disparity, got %d channels" % c1)
        c = ic.shape[1]
        nb = _lib.load().dn_edge_smooth_blocks(b, h, w)
        partial = torch.empty((nb, 2), dtype=torch.float32, device=dc.device)
        loss = torch.empty((), dtype=torch.float32, device=dc.device)
        _lib.call("dn_edge_smooth_fwd", dc.data_ptr(), ic.data_ptr(), b, c, h, w, partial.data_ptr(), loss.data_ptr(), _stream())
        ctx.save_for_backward(dc, ic)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        dc, ic = ctx.saved_tensors
        b, _, h, w = dc.shape
        dl = dloss.contiguous().float()
        g = torch.empty_like(dc)
        _lib.call("dn_edge_smooth_bwd", dc.data_ptr(), ic.data_ptr(), dl.data_ptr(), b, ic.shape[1], h, w, g.data_ptr(), _stream())
        return g, None


def get_smooth_loss(disp, img):
    """reference layers.py:199-212: edge-aware first-order smoothness (gradient w.r.t. the disparity)."""
    return _EdgeSmooth.apply(disp, img)


def upsample(x):
    """reference layers.py:193-196 (nearest x2).  Inside networks.DepthDecoder the upsample is virtual (the consumer conv's
    loader reads index >> 1); this standalone form is kept for API completeness and is not on the hot path."""
    raise NotImplementedError("layers.upsample is fused into the consumer convolution on the HIP path (networks.DepthDecoder)")


class Conv3x3(nn.Module):
    """reference layers.py:124-136: ReflectionPad2d(1) (or ZeroPad2d) + Conv2d(3x3).  Parameter container: executed by
    networks.DepthDecoder through the engine (reflection is a loader index map, not a padded copy)."""

    def __init__(self, in_channels, out_channels, use_refl=True):
        super(Conv3x3, self).__init__()
        self.pad = nn.ReflectionPad2d(1) if use_refl else nn.ZeroPad2d(1)
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), 3)


class ConvBlock(nn.Module):
    """reference layers.py:106-121: Conv3x3 + ELU (parameter container, see Conv3x3)."""

    def __init__(self, in_channels, out_channels):
        super(ConvBlock, self).__init__()
        self.conv = Conv3x3(in_channels, out_channels)
        self.nonlin = nn.ELU(inplace=True)
