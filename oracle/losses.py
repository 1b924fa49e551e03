"""Oracle (torch-CPU) restatement of the reference's per-pixel depth losses and metrics.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Reference: loss_functions.py.
All functions take/return torch tensors and are differentiable where the reference is.
Deliberately replicated reference quirks are marked QUIRK.
"""
import torch
import torch.nn.functional as F

from .geometry import inverse_warp

MAX_DEPTH = {"kitti": 80.0, "nyu": 10.0}


def _max_depth(datasets):
    if datasets not in MAX_DEPTH:
        raise ValueError("undefined datasets %r" % (datasets,))
    return MAX_DEPTH[datasets]


def _valid_pairs(gt, pred, max_depth):
    """(gt[valid], clamp(pred[valid], 1e-3, max)) with valid = 0 < gt < max.
    loss_functions.py:111-115 and the same pattern in every supervised loss."""
    valid = (gt > 0) & (gt < max_depth)
    return gt[valid], pred[valid].clamp(1e-3, max_depth), valid


# ---------------------------------------------------------------- per-sample losses (scale 0)
def _per_sample(gt_depth, depth, datasets, fn):
    pred = depth[0][:, 0]
    mx = _max_depth(datasets)
    total = 0
    for g, p in zip(gt_depth, pred):
        vg, vp, valid = _valid_pairs(g, p, mx)
        total = total + fn(vg, vp, valid)
    return total / pred.shape[0]


def l1_loss(gt_depth, depth, datasets):
    """loss_functions.py:104-129.  Mean over valid pixels per sample, then mean over the batch;
    an empty mask yields NaN (mean of empty), like the reference."""
    return _per_sample(gt_depth, depth, datasets, lambda g, p, v: (g - p).abs().mean())


def l2_loss(gt_depth, depth, datasets):
    """loss_functions.py:77-102.  QUIRK: the 'nyu' branch is an L1 (line 97)."""
    if datasets == "nyu":
        return l1_loss(gt_depth, depth, datasets)
    return _per_sample(gt_depth, depth, datasets, lambda g, p, v: ((g - p) ** 2).mean())


def _berhu_term(g, p):
    # loss_functions.py:142-147
    r = (g - p).abs()
    c = 0.2 * r.max()
    return torch.where(r > c, (r ** 2 + c ** 2) / (2 * c), r).mean()


def berhu_loss(gt_depth, depth, datasets):
    """loss_functions.py:131-161.  The reference's 'nyu' branch lacks its for-loop (NameError);
    here it loops like the 'kitti' branch with max depth 10."""
    return _per_sample(gt_depth, depth, datasets, lambda g, p, v: _berhu_term(g, p))


def _scale_inv_term(g, p, valid):
    # loss_functions.py:173-177
    n = valid.sum().to(torch.float32)
    return ((g.abs() - p.abs()) ** 2).mean() - 0.5 * (g - p).sum() ** 2 / (n ** 2)


def Scale_invariant_loss(gt_depth, depth, datasets):
    """loss_functions.py:163-189."""
    return _per_sample(gt_depth, depth, datasets, _scale_inv_term)


# --------------------------------------------------------------------------- pyramids
def generate_max_pyramid(image):
    pyr = [image]
    for _ in range(3):
        pyr.append(F.max_pool2d(pyr[-1], 2, 2))
    return pyr


def generate_avg_pyramid(image):
    pyr = [image]
    for _ in range(3):
        pyr.append(F.avg_pool2d(pyr[-1], 2, 2))
    return pyr


def generate_bilinear_pyramid(image):
    """loss_functions.py:205-215: bilinear x0.5 (align_corners=False) of the sparse GT, 3 times."""
    pyr, cur = [image], image.unsqueeze(1)
    for _ in range(3):
        cur = F.interpolate(cur, scale_factor=0.5, mode="bilinear", align_corners=False)
        pyr.append(cur.squeeze())
    return pyr


def _gt_pyramid(gt_depth, pool_type):
    if pool_type == "max":
        return generate_max_pyramid(gt_depth)
    if pool_type == "avg":
        return generate_avg_pyramid(gt_depth)
    if pool_type == "bilinear":
        return generate_bilinear_pyramid(gt_depth)
    raise ValueError("undefined pool type")


def _multiscale(gt_list, depth, fn):
    """One mask over the WHOLE batch per scale, weight 1/2^i, max depth hard-coded 80
    (loss_functions.py:229-238)."""
    total = 0
    for i, d in enumerate(depth):
        g, p, valid = _valid_pairs(gt_list[i], d.squeeze(), 80.0)
        total = total + fn(g, p, valid) / (2 ** i)
    return total


def Multiscale_L1_loss(gt_depth, depth, pool_type="bilinear"):
    """loss_functions.py:217-238."""
    return _multiscale(_gt_pyramid(gt_depth, pool_type), depth, lambda g, p, v: (g - p).abs().mean())


def Multiscale_FULL_L1_loss(gt_depth, depth, pool_type="bilinear"):
    """loss_functions.py:240-257: predictions upsampled to full resolution instead."""
    up = [F.interpolate(d, scale_factor=2 ** i, mode=pool_type) for i, d in enumerate(depth)]
    return _multiscale([gt_depth] * len(depth), up, lambda g, p, v: (g - p).abs().mean())


def Multiscale_L2_loss(gt_depth, depth):
    """loss_functions.py:259-273."""
    return _multiscale(generate_bilinear_pyramid(gt_depth), depth, lambda g, p, v: ((g - p) ** 2).mean())


def Multiscale_berhu_loss(gt_depth, depth):
    """loss_functions.py:275-296."""
    return _multiscale(generate_bilinear_pyramid(gt_depth), depth, lambda g, p, v: _berhu_term(g, p))


def Multiscale_scale_inv_loss(gt_depth, depth):
    """loss_functions.py:298-315."""
    return _multiscale(generate_bilinear_pyramid(gt_depth), depth, _scale_inv_term)


# ------------------------------------------------------------------------- smoothness
def _second_order_abs_means(m):
    dy = m[:, :, 1:] - m[:, :, :-1]
    dx = m[:, :, :, 1:] - m[:, :, :, :-1]
    dx2 = dx[:, :, :, 1:] - dx[:, :, :, :-1]
    dxdy = dx[:, :, 1:] - dx[:, :, :-1]
    dydx = dy[:, :, :, 1:] - dy[:, :, :, :-1]
    dy2 = dy[:, :, 1:] - dy[:, :, :-1]
    return dx2.abs().mean() + dxdy.abs().mean() + dydx.abs().mean() + dy2.abs().mean()


def smooth_loss(pred_map):
    """loss_functions.py:367-386: second-order differences, scale weights 1, 1/2.3, 1/2.3^2 ..."""
    if not isinstance(pred_map, (tuple, list)):
        pred_map = [pred_map]
    total, weight = 0, 1.0
    for m in pred_map:
        total = total + _second_order_abs_means(m) * weight
        weight /= 2.3
    return total


def smooth_DORN_loss(pred_map):
    """loss_functions.py:388-399."""
    return _second_order_abs_means(pred_map)


# ------------------------------------------------------------------------------ DORN
def DORN_loss(gt_depth, ord_labels, target, datasets):
    """loss_functions.py:16-74.  ord_labels [N,K,H,W] = P(k-th threshold passed); target [N,H,W] int.
    loss = -( sum_{k<=t-1, valid} log clamp(P_k) + sum_{k>t-1, valid} log clamp(1-P_k) ) / n_valid_pixels."""
    mx = _max_depth(datasets)
    n, k, h, w = ord_labels.shape
    valid = ((gt_depth > 0) & (gt_depth < mx)).unsqueeze(1)
    kk = torch.arange(k, dtype=torch.int32).view(1, k, 1, 1)
    t = target.unsqueeze(1).to(torch.int32)
    le = (kk <= t - 1) & valid
    gt_ = (kk > t - 1) & valid
    s = torch.log(ord_labels[le].clamp(1e-8, 1e8)).sum() + torch.log((1 - ord_labels[gt_]).clamp(1e-8, 1e8)).sum()
    return s / (-(valid.sum().to(torch.float32)))


# ------------------------------------------------------------------ photometric / masks
def explainability_loss(mask):
    """loss_functions.py:357-364: BCE against all-ones, summed over scales."""
    if not isinstance(mask, (tuple, list)):
        mask = [mask]
    total = 0
    for m in mask:
        total = total + F.binary_cross_entropy(m, torch.ones_like(m))
    return total


def photometric_reconstruction_loss(tgt_img, ref_imgs, intrinsics, intrinsics_inv, depth, explainability_mask,
                                    pose, rotation_mode="euler", padding_mode="zeros", align_corners=False):
    """loss_functions.py:317-354.  Per scale: area-downsample images, rescale K, inverse-warp every ref,
    out-of-bound mask = 1 - prod_c(warped_c == 0), |diff|.mean(); summed over refs and scales.
    `align_corners` is the grid_sample flag (reference leaves torch's default = False on torch>=1.3)."""
    if not isinstance(explainability_mask, (tuple, list)):
        explainability_mask = [explainability_mask]
    if not isinstance(depth, (tuple, list)):
        depth = [depth]
    total = 0
    for d, mask in zip(depth, explainability_mask):
        b, _, h, w = d.shape
        down = tgt_img.shape[2] / h
        tgt_s = F.interpolate(tgt_img, (h, w), mode="area")
        refs_s = [F.interpolate(r, (h, w), mode="area") for r in ref_imgs]
        k_s = torch.cat((intrinsics[:, 0:2] / down, intrinsics[:, 2:]), dim=1)
        kinv_s = torch.cat((intrinsics_inv[:, :, 0:2] * down, intrinsics_inv[:, :, 2:]), dim=2)
        for i, ref in enumerate(refs_s):
            warped = inverse_warp(ref, d[:, 0], pose[:, i], k_s, kinv_s, rotation_mode, padding_mode, align_corners)
            oob = 1 - (warped == 0).prod(1, keepdim=True).type_as(warped)
            diff = (tgt_s - warped) * oob
            if mask is not None:
                diff = diff * mask[:, i:i + 1].expand_as(diff)
            total = total + diff.abs().mean()
    return total


# ---------------------------------------------------------------------------- metrics
def garg_crop_bounds(h, w):
    """loss_functions.py:414-415 (int() truncation): rows y1:y2, cols x1:x2."""
    return int(0.40810811 * h), int(0.99189189 * h), int(0.03594771 * w), int(0.96405229 * w)


@torch.no_grad()
def compute_errors(gt, pred, dataset="kitti", crop=True, unsupervised=False):
    """loss_functions.py:401-448 -> [abs_diff, abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3], batch means."""
    b, h, w = gt.shape
    if dataset == "kitti":
        mx = 80.0
        crop_mask = torch.zeros(h, w, dtype=torch.bool)
        if crop:
            y1, y2, x1, x2 = garg_crop_bounds(h, w)
            crop_mask[y1:y2, x1:x2] = True
    else:
        mx = 10.0
        crop_mask = torch.ones(h, w, dtype=torch.bool)
    acc = [0.0] * 8
    for g, p in zip(gt, pred):
        valid = (g > 0) & (g < mx)
        if crop:
            valid = valid & crop_mask
        vg = g[valid]
        vp = p[valid].clamp(1e-3, mx)
        if unsupervised:
            vp = vp * torch.median(vg) / torch.median(vp)
        thr = torch.max(vg / vp, vp / vg)
        terms = [
            (vg - vp).abs().mean(),
            ((vg - vp).abs() / vg).mean(),
            (((vg - vp) ** 2) / vg).mean(),
            torch.sqrt(((vg - vp) ** 2).mean()),
            torch.sqrt(((torch.log(vg) - torch.log(vp)) ** 2).mean()),
            (thr < 1.25).float().mean(),
            (thr < 1.25 ** 2).float().mean(),
            (thr < 1.25 ** 3).float().mean(),
        ]
        acc = [a + t for a, t in zip(acc, terms)]
    return [float(a) / b for a in acc]
