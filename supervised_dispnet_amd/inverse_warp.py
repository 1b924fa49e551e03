"""Mirror of the reference's inverse_warp.py on HIP kernels (same function names / arguments).

inverse_warp() is one fused kernel per direction (pixel2cam -> projection -> cam2pixel -> bilinear grid_sample) with an
analytic backward to depth and pose; there is no module-global pixel grid (the reference's `pixel_coords` cache,
inverse_warp.py:5-15, is a thread-safety hazard -- coordinates are generated in registers).
Gradients are provided w.r.t. `depth` and `pose` (what the photometric loss trains); `img` and the intrinsics are data.
"""
import torch

from . import _lib
from .engine import _stream, require_cuda

_ROT = {"euler": 0, "quat": 1}
_PAD = {"zeros": 0, "border": 1}


def check_sizes(input, input_name, expected):
    """reference inverse_warp.py:18-23"""
    condition = [input.ndimension() == len(expected)]
    for i, size in enumerate(expected):
        if size.isdigit():
            condition.append(input.size(i) == int(size))
    assert all(condition), "wrong size for {}, expected {}, got  {}".format(input_name, 'x'.join(expected), list(input.size()))


class _InverseWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, depth, pose, intrinsics, intrinsics_inv, rot, pad, align):
        for t, n in ((img, "img"), (depth, "depth"), (pose, "pose"), (intrinsics, "intrinsics"), (intrinsics_inv, "intrinsics_inv")):
            require_cuda(t, n)
        imgc, dc, pc = img.contiguous().float(), depth.contiguous().float(), pose.contiguous().float()
        K, Kinv = intrinsics.contiguous().float(), intrinsics_inv.contiguous().float()
        B, _, h, w = imgc.shape
        dev = imgc.device
        proj = torch.empty((B, 12), dtype=torch.float32, device=dev)
        kinv_s = torch.empty((B, 9), dtype=torch.float32, device=dev)
        _lib.call("dn_pose_proj_fwd", pc.data_ptr(), 6, K.data_ptr(), Kinv.data_ptr(), B, rot, 1.0, proj.data_ptr(), kinv_s.data_ptr(), _stream())
        out = torch.empty((B, 3, h, w), dtype=torch.float32, device=dev)
        _lib.call("dn_inverse_warp_fwd", imgc.data_ptr(), dc.data_ptr(), proj.data_ptr(), kinv_s.data_ptr(), B, h, w, pad, align,
                  out.data_ptr(), _stream())
        ctx.save_for_backward(imgc, dc, pc, K, proj, kinv_s)
        ctx.cfg = (B, h, w, rot, pad, align)
        return out

    @staticmethod
    def backward(ctx, dout):
        imgc, dc, pc, K, proj, kinv_s = ctx.saved_tensors
        B, h, w, rot, pad, align = ctx.cfg
        dev = imgc.device
        g = dout.contiguous().float()
        nb = _lib.load().dn_warp_blocks(h, w)
        ddepth = torch.empty_like(dc)
        dpp = torch.empty((B, nb, 12), dtype=torch.float32, device=dev)
        _lib.call("dn_inverse_warp_bwd", imgc.data_ptr(), dc.data_ptr(), proj.data_ptr(), kinv_s.data_ptr(), B, h, w, pad, align,
                  g.data_ptr(), ddepth.data_ptr(), 0, dpp.data_ptr(), _stream())
        dpose = torch.empty_like(pc)
        _lib.call("dn_pose_proj_bwd", pc.data_ptr(), 6, K.data_ptr(), B, rot, 1.0, dpp.data_ptr(), nb, dpose.data_ptr(), 6, 0, _stream())
        return None, ddepth, dpose, None, None, None, None, None


def inverse_warp(img, depth, pose, intrinsics, intrinsics_inv, rotation_mode='euler', padding_mode='zeros', align_corners=False):
    """reference inverse_warp.py:160-193.  `align_corners` is F.grid_sample's flag: the reference passes none, i.e. False on
    torch >= 1.3 and True on its pinned torch 1.0.1; both behaviours are pinned by golden vectors."""
    check_sizes(img, 'img', 'B3HW')
    check_sizes(depth, 'depth', 'BHW')
    check_sizes(pose, 'pose', 'B6')
    check_sizes(intrinsics, 'intrinsics', 'B33')
    check_sizes(intrinsics_inv, 'intrinsics', 'B33')
    assert intrinsics_inv.size() == intrinsics.size()
    return _InverseWarp.apply(img, depth, pose, intrinsics, intrinsics_inv, _ROT[rotation_mode], _PAD[padding_mode],
                              1 if align_corners else 0)


class _PoseVec2Mat(torch.autograd.Function):
    """[B,6] -> [B,3,4] = [R|t] with the analytic backward the warp kernels use (dn_pose_proj_fwd / dn_pose_proj_bwd with K = I)."""

    @staticmethod
    def forward(ctx, vec, rot):
        require_cuda(vec, "pose vector")
        v = vec.contiguous().float()
        B = v.shape[0]
        eye = torch.eye(3, dtype=torch.float32, device=v.device).repeat(B, 1, 1)
        proj = torch.empty((B, 12), dtype=torch.float32, device=v.device)
        scratch = torch.empty((B, 9), dtype=torch.float32, device=v.device)
        _lib.call("dn_pose_proj_fwd", v.data_ptr(), 6, eye.data_ptr(), eye.data_ptr(), B, rot, 1.0, proj.data_ptr(), scratch.data_ptr(), _stream())
        ctx.save_for_backward(v, eye)
        ctx.rot = rot
        return proj.view(B, 3, 4)

    @staticmethod
    def backward(ctx, dmat):
        v, eye = ctx.saved_tensors
        B = v.shape[0]
        dpp = dmat.contiguous().float().view(B, 1, 12)               # one "block" of projection-matrix gradients per sample
        dpose = torch.empty_like(v)
        _lib.call("dn_pose_proj_bwd", v.data_ptr(), 6, eye.data_ptr(), B, ctx.rot, 1.0, dpp.data_ptr(), 1, dpose.data_ptr(), 6, 0, _stream())
        return dpose, None


def pose_vec2mat(vec, rotation_mode='euler'):
    """reference inverse_warp.py:141-157: [B,6] (tx,ty,tz,rx,ry,rz) -> [B,3,4] = [R|t], differentiable w.r.t. `vec` like the
    reference's torch expression."""
    return _PoseVec2Mat.apply(vec, _ROT[rotation_mode])


def euler2mat(angle):
    """reference inverse_warp.py:77-114: [B,3] -> [B,3,3] = Rx @ Ry @ Rz (differentiable)."""
    vec = torch.cat([torch.zeros_like(angle), angle], dim=1)
    return pose_vec2mat(vec, 'euler')[:, :, :3].contiguous()


def quat2mat(quat):
    """reference inverse_warp.py:117-138 (differentiable)."""
    vec = torch.cat([torch.zeros_like(quat), quat], dim=1)
    return pose_vec2mat(vec, 'quat')[:, :, :3].contiguous()
