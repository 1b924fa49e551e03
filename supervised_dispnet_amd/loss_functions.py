"""Mirror of the reference's loss_functions.py on HIP kernels (same function names, argument order and meaning).

Implemented on the MI355X path: l1_loss, l2_loss (incl. the reference's 'nyu' == L1 quirk), smooth_loss,
smooth_DORN_loss, compute_errors.  The remaining reference losses raise NotImplementedError until their kernels land
(no silent PyTorch fallback).
"""
import torch

from . import _lib
from .engine import _stream, require_cuda

_MAX_DEPTH = {"kitti": 80.0, "nyu": 10.0}


def _max_depth(datasets):
    if datasets not in _MAX_DEPTH:
        raise ValueError("undefined datasets %r" % (datasets,))
    return _MAX_DEPTH[datasets]


class _MaskedLoss(torch.autograd.Function):
    """loss_functions.py:77-129: per-sample mean over valid pixels of f(gt - clamp(pred, 1e-3, max)), batch mean."""

    @staticmethod
    def forward(ctx, gt, pred, max_depth, kind):
        require_cuda(pred, "predicted depth")
        require_cuda(gt, "ground-truth depth")
        B = pred.shape[0]
        gtc, pc = gt.contiguous().float(), pred.contiguous()
        pixels = pc.numel() // B
        if gtc.numel() != pc.numel():
            raise ValueError("gt %s and prediction %s sizes differ" % (tuple(gt.shape), tuple(pred.shape)))
        stats = torch.empty((B, 2), dtype=torch.float32, device=pc.device)
        loss = torch.empty((), dtype=torch.float32, device=pc.device)
        _lib.call("dn_masked_loss_fwd", gtc.data_ptr(), pc.data_ptr(), B, pixels, max_depth, kind, stats.data_ptr(),
                  loss.data_ptr(), _stream())
        ctx.save_for_backward(gtc, pc, stats)
        ctx.cfg = (B, pixels, max_depth, kind, pred.shape)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        gtc, pc, stats = ctx.saved_tensors
        B, pixels, max_depth, kind, shape = ctx.cfg
        dl = dloss.contiguous().float()
        dpred = torch.empty(shape, dtype=torch.float32, device=pc.device)
        _lib.call("dn_masked_loss_bwd", gtc.data_ptr(), pc.data_ptr(), stats.data_ptr(), dl.data_ptr(), B, pixels,
                  max_depth, kind, dpred.data_ptr(), _stream())
        return None, dpred, None, None


def _scale0(depth):
    """depth[0][:,0] of the reference, handed over as the whole [B,1,H,W] tensor (same memory, no select node)."""
    d0 = depth[0]
    if d0.dim() != 4 or d0.shape[1] != 1:
        raise ValueError("expected depth[0] of shape [B,1,H,W], got %s" % (tuple(d0.shape),))
    return d0


def l1_loss(gt_depth, depth, datasets):
    """reference loss_functions.py:104-129 (uses scale 0 only: depth[0][:,0])."""
    return _MaskedLoss.apply(gt_depth, _scale0(depth), _max_depth(datasets), _lib.LOSS_L1)


def l2_loss(gt_depth, depth, datasets):
    """reference loss_functions.py:77-102; its 'nyu' branch is an L1 (line 97) and is reproduced as such."""
    kind = _lib.LOSS_L1 if datasets == "nyu" else _lib.LOSS_L2
    return _MaskedLoss.apply(gt_depth, _scale0(depth), _max_depth(datasets), kind)


class _Smooth(torch.autograd.Function):
    """loss_functions.py:367-386 for a list of [B,C,H,W] maps (each channel treated as its own map)."""

    @staticmethod
    def forward(ctx, weight_decay, *maps):
        dev = maps[0].device
        loss = torch.zeros((), dtype=torch.float32, device=dev)
        saved, cfg, w = [], [], 1.0
        lib = _lib.load()
        for m in maps:
            require_cuda(m, "prediction map")
            mc = m.contiguous()
            b, c, h, wd = mc.shape
            nb = lib.dn_smooth_blocks(b * c, h, wd)
            partial = torch.empty((nb, 4), dtype=torch.float32, device=dev)
            _lib.call("dn_smooth2_fwd", mc.data_ptr(), b * c, h, wd, w, partial.data_ptr(), loss.data_ptr(), _stream())
            saved.append(mc)
            cfg.append((b * c, h, wd, w))
            w /= weight_decay
        ctx.save_for_backward(*saved)
        ctx.cfg = cfg
        return loss

    @staticmethod
    def backward(ctx, dloss):
        dl = dloss.contiguous().float()
        grads = []
        for mc, (bc, h, wd, w) in zip(ctx.saved_tensors, ctx.cfg):
            g = torch.empty_like(mc)
            _lib.call("dn_smooth2_bwd", mc.data_ptr(), dl.data_ptr(), bc, h, wd, w, g.data_ptr(), _stream())
            grads.append(g)
        return (None,) + tuple(grads)


def smooth_loss(pred_map):
    """reference loss_functions.py:367-386: second-order differences, per-scale weight 1, 1/2.3, 1/2.3^2, ..."""
    if type(pred_map) not in [tuple, list]:
        pred_map = [pred_map]
    return _Smooth.apply(2.3, *pred_map)


def smooth_DORN_loss(pred_map):
    """reference loss_functions.py:388-399 (single map, all K channels in one mean)."""
    return _Smooth.apply(2.3, pred_map)


@torch.no_grad()
def compute_errors(gt, pred, dataset='kitti', crop=True, unsupervised=False):
    """reference loss_functions.py:401-448 -> [abs_diff, abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3] (python floats)."""
    if unsupervised:
        raise NotImplementedError("median-scaled metrics (unsupervised=True) are not on the HIP path yet")
    require_cuda(pred, "predicted depth")
    b, h, w = gt.shape
    if dataset == 'kitti':
        mx = 80.0
        y1, y2, x1, x2 = (int(0.40810811 * h), int(0.99189189 * h), int(0.03594771 * w), int(0.96405229 * w)) if crop else (0, h, 0, w)
    else:
        mx = 10.0
        y1, y2, x1, x2 = 0, h, 0, w
    gtc, pc = gt.contiguous().float(), pred.contiguous().float()
    scratch = torch.empty((b, 9), dtype=torch.float32, device=pc.device)
    out = torch.empty(8, dtype=torch.float32, device=pc.device)
    _lib.call("dn_compute_errors", gtc.data_ptr(), pc.data_ptr(), b, h, w, mx, y1, y2, x1, x2, scratch.data_ptr(),
              out.data_ptr(), _stream())
    return out.tolist()


def _pending(name):
    def fn(*a, **k):
        raise NotImplementedError("loss_functions.%s has no HIP kernel yet in this build (no PyTorch fallback by design)" % name)
    fn.__name__ = name
    return fn


for _n in ("berhu_loss", "Scale_invariant_loss", "Multiscale_L1_loss", "Multiscale_FULL_L1_loss", "Multiscale_L2_loss",
           "Multiscale_berhu_loss", "Multiscale_scale_inv_loss", "photometric_reconstruction_loss", "explainability_loss",
           "DORN_loss"):
    globals()[_n] = _pending(_n)
