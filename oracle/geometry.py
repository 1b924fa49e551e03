"""Oracle (torch-CPU) restatement of the reference's inverse warp geometry.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Reference: inverse_warp.py:8-193.
No module-global pixel-grid cache (the reference's is keyed on H only, inverse_warp.py:5,36);
the grid is rebuilt per call, which gives the same values.
"""
import torch
import torch.nn.functional as F


def pixel2cam(depth, intrinsics_inv):
    """inverse_warp.py:26-40: cam = K^-1 [j, i, 1]^T * depth  ->  [B,3,H,W]."""
    b, h, w = depth.shape
    ii = torch.arange(h, dtype=depth.dtype).view(h, 1).expand(h, w)
    jj = torch.arange(w, dtype=depth.dtype).view(1, w).expand(h, w)
    pix = torch.stack((jj, ii, torch.ones_like(ii)), 0).reshape(1, 3, -1).expand(b, 3, h * w)
    return (intrinsics_inv @ pix).reshape(b, 3, h, w) * depth.unsqueeze(1)


def euler2mat(angle):
    """inverse_warp.py:77-114: R = Rx(x) @ Ry(y) @ Rz(z)."""
    x, y, z = angle[:, 0], angle[:, 1], angle[:, 2]
    zero = z.detach() * 0
    one = zero + 1
    cz, sz, cy, sy, cx, sx = torch.cos(z), torch.sin(z), torch.cos(y), torch.sin(y), torch.cos(x), torch.sin(x)
    bsz = angle.shape[0]
    zm = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], 1).reshape(bsz, 3, 3)
    ym = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], 1).reshape(bsz, 3, 3)
    xm = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], 1).reshape(bsz, 3, 3)
    return xm @ ym @ zm


def quat2mat(quat):
    """inverse_warp.py:117-138: (x,y,z) part of a quaternion with w := 1 before normalisation."""
    q = torch.cat([quat[:, :1].detach() * 0 + 1, quat], 1)
    q = q / q.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], 1).reshape(quat.shape[0], 3, 3)


def pose_vec2mat(vec, rotation_mode="euler"):
    """inverse_warp.py:141-157: vec = (tx,ty,tz,rx,ry,rz) -> [R|t]  [B,3,4]."""
    rot = euler2mat(vec[:, 3:]) if rotation_mode == "euler" else quat2mat(vec[:, 3:])
    return torch.cat([rot, vec[:, :3].unsqueeze(-1)], 2)


def cam2pixel(cam_coords, proj_rot, proj_tr, padding_mode):
    """inverse_warp.py:43-74: project, Z clamped at 1e-3, normalise to [-1,1] with the (w-1),(h-1)
    convention; padding 'zeros': any coordinate outside [-1,1] := 2 (no gradient through it)."""
    b, _, h, w = cam_coords.shape
    pc = proj_rot @ cam_coords.reshape(b, 3, -1) + proj_tr
    z = pc[:, 2].clamp(min=1e-3)
    xn = 2 * (pc[:, 0] / z) / (w - 1) - 1
    yn = 2 * (pc[:, 1] / z) / (h - 1) - 1
    if padding_mode == "zeros":
        xm = ((xn > 1) | (xn < -1)).detach()
        ym = ((yn > 1) | (yn < -1)).detach()
        xn = torch.where(xm, torch.full_like(xn, 2.0), xn)
        yn = torch.where(ym, torch.full_like(yn, 2.0), yn)
    return torch.stack([xn, yn], 2).reshape(b, h, w, 2)


def inverse_warp(img, depth, pose, intrinsics, intrinsics_inv, rotation_mode="euler", padding_mode="zeros",
                 align_corners=False):
    """inverse_warp.py:160-193.  `align_corners` is grid_sample's flag: the reference passes none, which
    means False on torch >= 1.3 (the oracle container) and meant True on its pinned torch 1.0.1
     -- both are pinned by goldens."""
    cam = pixel2cam(depth, intrinsics_inv)
    proj = intrinsics @ pose_vec2mat(pose, rotation_mode)
    grid = cam2pixel(cam, proj[:, :, :3], proj[:, :, -1:], padding_mode)
    return F.grid_sample(img, grid, padding_mode=padding_mode, align_corners=align_corners)
