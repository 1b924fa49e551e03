"""Small differentiable HIP ops used by the callers of the hot path (train.py) -- not part of the reference's module
namespace, but the reference's call sites have an exact counterpart here."""
import torch

from . import _lib
from .engine import _stream, require_cuda


class _Reciprocal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        require_cuda(x, "disparity")
        xc = x.contiguous()
        y = torch.empty_like(xc)
        _lib.call("dn_reciprocal_fwd", xc.data_ptr(), y.data_ptr(), xc.numel(), _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dyc = dy.contiguous()
        dx = torch.empty_like(y)
        _lib.call("dn_reciprocal_bwd", dyc.data_ptr(), y.data_ptr(), dx.data_ptr(), y.numel(), _stream())
        return dx


class _Holder:
    """Carries a tensor into an autograd.Function WITHOUT making it an input of the node."""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


class _ReciprocalFused(torch.autograd.Function):
    """depth = 1 / disp whose forward value the disparity head's kernel has already written (dn_conv_desc.recip_out): no launch in
    the forward pass; the backward is _Reciprocal's (d disp = -d depth * depth^2)."""

    @staticmethod
    def forward(ctx, x, holder):
        y = holder.t
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dyc = dy.contiguous()
        dx = torch.empty_like(y)
        _lib.call("dn_reciprocal_bwd", dyc.data_ptr(), y.data_ptr(), dx.data_ptr(), y.numel(), _stream())
        return dx, None


def reciprocal(disp):
    """depth = 1/disp  (reference train.py:445, `depth = [1/disp for disp in disparities]`).  A disparity that comes straight out of a
    network of this package carries its reciprocal with it (models/_common.run_net): the head kernel wrote both."""
    fused = getattr(disp, "_dn_recip", None)
    if fused is not None and fused[1] == disp._version and fused[0].shape == disp.shape:
        # handed out ONCE: the buffer becomes the caller's depth tensor, so a second reciprocal(disp) -- or one after an in-place
        # operation on the depth this call returned -- takes the plain kernel and gets a tensor of its own
        disp._dn_recip = None
        return _ReciprocalFused.apply(disp, _Holder(fused[0]))
    return _Reciprocal.apply(disp)


def disp_to_depth_list(disparities):
    if isinstance(disparities, (tuple, list)):
        return [reciprocal(d) for d in disparities]
    return reciprocal(disparities)
