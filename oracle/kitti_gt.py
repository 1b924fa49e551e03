"""Oracle (numpy) restatement of the KITTI Eigen ground-truth generation -- integer/index work,
must be bit-exact.  TEST INFRASTRUCTURE ONLY.

Reference: kitti_eval/depth_evaluation_utils.py: sub2ind 168-170, generate_depth_map 173-223,
generate_mask 236-248, generate_nyu_mask 250-254; test_disp.py compute_errors 453-469.
The file-reading half (calibration text, velodyne .bin) is replaced by array arguments.
"""
from collections import Counter

import numpy as np


def sub2ind(matrix_size, row_sub, col_sub):
    """QUIRK (kept bit-exactly): linearises with (n-1), not n, so distinct pixels can collide."""
    _, n = matrix_size
    return row_sub * (n - 1) + col_sub - 1


def project_velodyne(velo, p_rect, r_rect, velo2cam_rt, im_shape):
    """velo [N,4] (x fwd, y left, z up, reflectance=1) -> in-image points [M,3] = (col, row, depth).
    Steps and float64 arithmetic of generate_depth_map lines 175-201."""
    velo2cam = np.vstack((velo2cam_rt, np.array([0, 0, 0, 1.0])))
    r4 = np.eye(4)
    r4[:3, :3] = r_rect.reshape(3, 3)
    p_velo2im = np.dot(np.dot(p_rect.reshape(3, 4), r4), velo2cam)
    velo = velo[velo[:, 0] >= 0, :]
    pts = np.dot(p_velo2im, velo.T).T
    pts[:, :2] = pts[:, :2] / pts[:, -1:]
    pts[:, 0] = np.round(pts[:, 0]) - 1
    pts[:, 1] = np.round(pts[:, 1]) - 1
    ok = (pts[:, 0] >= 0) & (pts[:, 1] >= 0)
    ok = ok & (pts[:, 0] < im_shape[1]) & (pts[:, 1] < im_shape[0])
    return pts[ok, :]


def scatter_depth_min_duplicates(pts, im_shape):
    """Lines 203-215: last-write-wins scatter, then every colliding sub2ind key takes the MIN depth of
    its group, written at the pixel of the group's first point."""
    depth = np.zeros(im_shape)
    depth[pts[:, 1].astype(int), pts[:, 0].astype(int)] = pts[:, 2]
    inds = sub2ind(depth.shape, pts[:, 1], pts[:, 0])
    dupes = [item for item, count in Counter(inds).items() if count > 1]
    for dd in dupes:
        where = np.where(inds == dd)[0]
        depth[int(pts[where[0], 1]), int(pts[where[0], 0])] = pts[where, 2].min()
    depth[depth < 0] = 0
    return depth


def generate_depth_map(velo, p_rect, r_rect, velo2cam_rt, im_shape):
    return scatter_depth_min_duplicates(project_velodyne(velo, p_rect, r_rect, velo2cam_rt, im_shape), im_shape)


def garg_crop(gt_height, gt_width):
    """Lines 242-243: float products truncated by astype(int32)."""
    return np.array([0.40810811 * gt_height, 0.99189189 * gt_height,
                     0.03594771 * gt_width, 0.96405229 * gt_width]).astype(np.int32)


def generate_mask(gt_depth, min_depth, max_depth):
    mask = np.logical_and(gt_depth > min_depth, gt_depth < max_depth)
    c = garg_crop(*gt_depth.shape)
    crop_mask = np.zeros(mask.shape)
    crop_mask[c[0]:c[1], c[2]:c[3]] = 1
    return np.logical_and(mask, crop_mask)


def generate_nyu_mask(gt_depth, min_depth, max_depth):
    return np.logical_and(gt_depth > min_depth, gt_depth < max_depth)


def compute_errors_np(gt, pred):
    """test_disp.py:453-469 -> abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3 (numpy, 1-D valid arrays)."""
    thresh = np.maximum(gt / pred, pred / gt)
    a1 = (thresh < 1.25).mean()
    a2 = (thresh < 1.25 ** 2).mean()
    a3 = (thresh < 1.25 ** 3).mean()
    rmse = np.sqrt(((gt - pred) ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    abs_rel = np.mean(np.abs(gt - pred) / gt)
    sq_rel = np.mean(((gt - pred) ** 2) / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3


def compute_abs_rel_per_pixel(gt, pred, min_depth, max_depth):
    """test_disp.py:471-477."""
    valid = (gt > min_depth) & (gt < max_depth)
    valid_complement = np.logical_not(valid)
    with np.errstate(divide="ignore", invalid="ignore"):
        abs_rel = np.abs(gt - pred) / gt
    abs_rel[valid_complement] = -1
    return abs_rel


def worst_pixels_loop(abs_rel_map, count=300):
    """test_disp.py:318-338 restated WITHOUT numpy's selection routine: every valid pixel inside the Garg crop in row-major order,
    then the `count` largest abs_rel values by a full stable sort.  Returns the selected (row, col) pairs as a sorted list of tuples --
    the SET the reference's np.argpartition(..., -300)[-300:] selects whenever the 300th and 301st largest values differ (the
    order inside argpartition's output is an implementation detail of numpy's introselect and is compared separately, through the
    committed golden)."""
    h, w = abs_rel_map.shape
    c = garg_crop(h, w)
    rows = []
    for y in range(int(c[0]), int(c[1])):
        for x in range(int(c[2]), int(c[3])):
            v = abs_rel_map[y, x]
            if v > 0:
                rows.append((float(v), y, x))
    rows.sort(key=lambda r: r[0])
    return sorted((y, x) for _, y, x in rows[-count:]), (rows[-count][0] if len(rows) >= count else None), \
        (rows[-count - 1][0] if len(rows) > count else None)
