"""KITTI Eigen-split ground truth and evaluation helpers for test_disp.py (host side, numpy).

Integer / index work -- results are bit-identical to kitti_eval/depth_evaluation_utils.py, including its quirks:
  * `sub2ind` linearises with (n-1) instead of n (:168-170), so DISTINCT pixels can share a duplicate key;
  * the scatter at :205 is last-write-wins, then every duplicate key writes the MIN depth of its group at the pixel of
    the group's FIRST point (:208-214), processed in order of first occurrence of the key.
The reference does this with a Counter and a Python loop per duplicate (the slow part of evaluation); here it is a stable
sort + segmented minimum, no Python loop over points.
"""
import os

import numpy as np


def read_calib_file(path):
    """key: v0 v1 ... lines -> dict of float arrays (strings kept when not numeric), :147-165."""
    float_chars = set("0123456789.e+- ")
    out = {}
    with open(path, "r") as f:
        for line in f:
            if ":" not in line:
                continue
            key, value = line.split(":", 1)
            value = value.strip()
            out[key] = value
            if float_chars.issuperset(value):
                try:
                    out[key] = np.array([float(v) for v in value.split(" ")])
                except ValueError:
                    pass
    return out


def load_velodyne_points(file_name):
    pts = np.fromfile(file_name, dtype=np.float32).reshape(-1, 4)
    pts[:, 3] = 1
    return pts


def project_velodyne(velo, p_rect, r_rect, velo2cam_rt, im_shape):
    """velo [N,4] -> in-image points [M,3] = (col, row, depth), float64, :175-201."""
    velo2cam = np.vstack((np.asarray(velo2cam_rt, dtype=np.float64), np.array([0, 0, 0, 1.0])))
    r4 = np.eye(4)
    r4[:3, :3] = np.asarray(r_rect, dtype=np.float64).reshape(3, 3)
    p_velo2im = np.dot(np.dot(np.asarray(p_rect, dtype=np.float64).reshape(3, 4), r4), velo2cam)
    velo = velo[velo[:, 0] >= 0, :]
    pts = np.dot(p_velo2im, velo.T).T
    pts[:, :2] = pts[:, :2] / pts[:, -1:]
    pts[:, 0] = np.round(pts[:, 0]) - 1
    pts[:, 1] = np.round(pts[:, 1]) - 1
    ok = (pts[:, 0] >= 0) & (pts[:, 1] >= 0) & (pts[:, 0] < im_shape[1]) & (pts[:, 1] < im_shape[0])
    return pts[ok, :]


def scatter_depth_min_duplicates(pts, im_shape):
    """:203-215 without the per-duplicate Python loop."""
    h, w = im_shape
    depth = np.zeros((h, w))
    rows, cols = pts[:, 1].astype(np.int64), pts[:, 0].astype(np.int64)
    depth[rows, cols] = pts[:, 2]                                   # last write wins
    if len(pts):
        key = rows * (w - 1) + cols - 1                              # the reference's (n-1) linearisation
        order = np.argsort(key, kind="stable")
        ks = key[order]
        start = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
        count = np.diff(np.r_[start, len(ks)])
        gmin = np.minimum.reduceat(pts[order, 2], start)
        first = order[start]                                         # stable sort: first point of each group
        dup = count > 1
        # the reference visits duplicate keys in order of first occurrence; targets can coincide across keys
        # (different keys, same first pixel is impossible, but keep the order anyway for exactness)
        visit = np.argsort(first[dup], kind="stable")
        fr, fc, fm = rows[first[dup]][visit], cols[first[dup]][visit], gmin[dup][visit]
        depth[fr, fc] = fm
    depth[depth < 0] = 0
    return depth


def generate_depth_map(calib_dir, velo_file_name, im_shape, cam=2):
    cam2cam = read_calib_file(os.path.join(calib_dir, "calib_cam_to_cam.txt"))
    velo2cam = read_calib_file(os.path.join(calib_dir, "calib_velo_to_cam.txt"))
    rt = np.hstack((velo2cam["R"].reshape(3, 3), velo2cam["T"][..., np.newaxis]))
    pts = project_velodyne(load_velodyne_points(velo_file_name), cam2cam["P_rect_0" + str(cam)], cam2cam["R_rect_00"], rt, im_shape)
    return scatter_depth_min_duplicates(pts, tuple(im_shape))


def garg_crop(gt_height, gt_width):
    return np.array([0.40810811 * gt_height, 0.99189189 * gt_height, 0.03594771 * gt_width, 0.96405229 * gt_width]).astype(np.int32)


def generate_mask(gt_depth, min_depth, max_depth):
    """valid-range mask AND the Garg ECCV16 crop (:236-248)."""
    mask = np.logical_and(gt_depth > min_depth, gt_depth < max_depth)
    c = garg_crop(*gt_depth.shape)
    crop = np.zeros(mask.shape, dtype=bool)
    crop[c[0]:c[1], c[2]:c[3]] = True
    return np.logical_and(mask, crop)


def generate_nyu_mask(gt_depth, min_depth, max_depth):
    return np.logical_and(gt_depth > min_depth, gt_depth < max_depth)


def compute_errors(gt, pred):
    """test_disp.py:453-469 -> abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3."""
    thresh = np.maximum(gt / pred, pred / gt)
    a1, a2, a3 = (thresh < 1.25).mean(), (thresh < 1.25 ** 2).mean(), (thresh < 1.25 ** 3).mean()
    rmse = np.sqrt(((gt - pred) ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    return np.mean(np.abs(gt - pred) / gt), np.mean(((gt - pred) ** 2) / gt), rmse, rmse_log, a1, a2, a3


def compute_abs_rel_per_pixel(gt, pred, min_depth, max_depth):
    """test_disp.py:471-477 -> |gt - pred| / gt per pixel, -1 where gt is outside (min_depth, max_depth).  (gt == 0 pixels divide by
    zero exactly like the reference and are then overwritten with -1.)"""
    valid = (gt > min_depth) & (gt < max_depth)
    with np.errstate(divide="ignore", invalid="ignore"):
        abs_rel = np.abs(gt - pred) / gt
    abs_rel[np.logical_not(valid)] = -1
    return abs_rel


def worst_pixels(abs_rel_map, count=300, threshold=None):
    """test_disp.py:318-343: the pixels with abs_rel > 0 inside the Garg crop, as (row, col, abs_rel) rows in np.where order; then
    either the `count` largest through np.argpartition(values, -count)[-count:] (the reference's `worst = True` branch, count 300)
    or, with `threshold`, the ones above it (the `else` branch, 0.3).  Returns (graph_index int32 [k, 2], index_result float64 [n, 3]).
    Index arithmetic -- identical to the reference's for identical inputs (same numpy selection routine on the same array)."""
    valid = abs_rel_map > 0
    gt_height, gt_width = valid.shape[:2]
    c = garg_crop(gt_height, gt_width)
    crop_mask = np.zeros(valid.shape)
    crop_mask[c[0]:c[1], c[2]:c[3]] = 1
    valid = np.logical_and(valid, crop_mask)
    ind = np.where(valid)
    index_result = np.zeros((len(ind[0]), 3))
    index_result[:, 0] = ind[0]
    index_result[:, 1] = ind[1]
    index_result[:, 2] = abs_rel_map[valid]
    if threshold is None:
        if len(ind[0]) < count:
            raise ValueError("only %d valid pixels inside the crop, %d requested (np.argpartition would raise too)" % (len(ind[0]), count))
        pick = np.argpartition(index_result[:, 2], -count)[-count:]
    else:
        pick = index_result[:, 2] > threshold
    return index_result[pick, :2].astype(np.int32), index_result


def annotate_pixels(tgt, graph_index, size=5):
    """test_disp.py:345-350: a size x size patch at every selected pixel is tinted red (R -> R/2 + 127.5, G, B -> /2, truncated),
    applied sequentially per patch offset on a copy of the uint8-valued image [H, W, 3]."""
    out = np.copy(tgt)
    for k in range(size):
        for l in range(size):
            y, x = graph_index[:, 0] + k, graph_index[:, 1] + l
            out[(y, x, 0)] = (out[(y, x, 0)] / 2.0 + 255.0 / 2.0).astype(int)
            out[(y, x, 1)] = (out[(y, x, 1)] / 2.0).astype(int)
            out[(y, x, 2)] = (out[(y, x, 2)] / 2.0).astype(int)
    return out


def imresize_bilinear(arr, size):
    """scipy.misc.imresize(arr, (h, w)) as test_disp.py:194 uses it (removed from SciPy): byte-scale the float image by its own
    min/max to uint8, PIL bilinear resize, back to an array."""
    from PIL import Image
    a = np.asarray(arr)
    cmin, cmax = float(a.min()), float(a.max())
    scale = 255.0 / (cmax - cmin) if cmax > cmin else 1.0
    b = ((a - cmin) * scale + 0.5).clip(0, 255).astype(np.uint8) if a.dtype != np.uint8 else a
    im = Image.fromarray(b)
    return np.asarray(im.resize((int(size[1]), int(size[0])), resample=Image.BILINEAR))


class KittiTestFramework(object):
    """test_framework_KITTI (:11-30) for a list of 'date/scene/image_0X/data/index.png' entries (no pose displacements:
    the PoseNet-scaled evaluation is outside this path)."""

    def __init__(self, root, test_files, min_depth=1e-3, max_depth=80):
        self.root, self.min_depth, self.max_depth = root, min_depth, max_depth
        self.items = []
        for sample in test_files:
            img = os.path.join(root, sample)
            date, scene, cam_id, _, index = sample[:-4].split("/")
            if not os.path.isfile(img):
                print("{} missing".format(img))
                continue
            vel = os.path.join(root, date, scene, "velodyne_points", "data", "{}.bin".format(index[:10]))
            self.items.append((img, os.path.join(root, date), vel, int(cam_id[-2:])))

    def __getitem__(self, i):
        from .data import load_as_float
        img, calib, vel, cam = self.items[i]
        tgt = load_as_float(img)
        depth = generate_depth_map(calib, vel, tgt.shape[:2], cam)
        return {"tgt": tgt, "path": img, "gt_depth": depth, "mask": generate_mask(depth, self.min_depth, self.max_depth)}

    def __len__(self):
        return len(self.items)


class NyuTestFramework(object):
    """test_framework_NYU (:33-66): root/nyu_depth_v2/labeled/npy/{images,depths}.npy."""

    def __init__(self, root, min_depth=1e-3, max_depth=10):
        folder = os.path.join(root, "nyu_depth_v2", "labeled", "npy")
        self.images = np.load(os.path.join(folder, "images.npy"))
        self.depths = np.load(os.path.join(folder, "depths.npy"))
        self.min_depth, self.max_depth = min_depth, max_depth

    def __getitem__(self, i):
        d = self.depths[i]
        return {"tgt": self.images[i], "gt_depth": d, "mask": generate_nyu_mask(d, self.min_depth, self.max_depth)}

    def __len__(self):
        return len(self.images)
