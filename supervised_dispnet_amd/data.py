"""Host-side sample sources for train.py / test_disp.py.

Only what the training hot path needs to be driven from the reference's on-disk layouts (the data pipeline itself is out of
scope): the scene-folder format of datasets/sequence_folders.py:14-76 and
datasets/validation_folders.py:28-60 (JPEG frames + per-frame .npy depth + cam.txt), the reference's transform chain
(custom_transforms.py: RandomHorizontalFlip 56-72, ArrayToTensor 40-53 with its /255, Normalize 25-37), and a synthetic
source with the statistics of SURVEY.md section 8d for boxes without a dataset.  JPEG decode is PIL (imageio is absent).
"""
import os
import random

import numpy as np
import torch
import torch.utils.data as data


def load_as_float(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.float32)


def normalization(imagenet=False, monodepth2=False):
    """(mean, std) of train.py:121-132 / test_disp.py:203-211."""
    if imagenet:
        return ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0]) if monodepth2 else ([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    return [0.5, 0.5, 0.5], [0.5, 0.5, 0.5]


class Transform(object):
    """flip (train only) -> CHW float /255 -> normalise, applied coherently to images, depth and intrinsics."""

    def __init__(self, mean, std, flip):
        self.mean = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(3, 1, 1)
        self.flip = flip

    def __call__(self, images, gt_depth, intrinsics):
        if self.flip and random.random() < 0.5:
            images = [np.ascontiguousarray(np.fliplr(im)) for im in images]
            gt_depth = np.ascontiguousarray(np.fliplr(gt_depth))
            if intrinsics is not None:
                intrinsics = np.copy(intrinsics)
                intrinsics[0, 2] = images[0].shape[1] - intrinsics[0, 2]
        tensors = [(torch.from_numpy(np.transpose(im, (2, 0, 1))).float() / 255 - self.mean) / self.std for im in images]
        return tensors, torch.from_numpy(np.ascontiguousarray(gt_depth)).float(), intrinsics


def _scenes(root, list_name):
    with open(os.path.join(root, list_name)) as f:
        return [os.path.join(root, line.strip()) for line in f if line.strip()]


def _files(folder, ext):
    return sorted(os.path.join(folder, n) for n in os.listdir(folder) if n.endswith(ext))


class SequenceFolder(data.Dataset):
    """root/scene/{0000000.jpg, 0000000.npy, ..., cam.txt}; train.txt / val.txt list the scenes.
    Yields (tgt_img, gt_depth) -- or (tgt_img, ref_imgs, intrinsics, intrinsics_inv, gt_depth) with `with_refs`, the
    5-tuple the reference's --unsupervised branch expects but its loader no longer returns."""

    def __init__(self, root, seed=None, train=True, sequence_length=3, transform=None, percentage=1, with_refs=False):
        np.random.seed(seed)
        random.seed(seed)
        self.scenes = _scenes(root, "train.txt" if train else "val.txt")
        self.transform = transform
        self.with_refs = with_refs
        demi = (sequence_length - 1) // 2
        shifts = [s for s in range(-demi, demi + 1) if s != 0]
        samples = []
        for scene in self.scenes:
            intr = np.genfromtxt(os.path.join(scene, "cam.txt")).astype(np.float32).reshape(3, 3)
            imgs, depth = _files(scene, ".jpg"), _files(scene, ".npy")
            if len(imgs) < sequence_length:
                continue
            for i in range(demi, len(imgs) - demi):
                samples.append({"intrinsics": intr, "tgt": imgs[i], "ref_imgs": [imgs[i + s] for s in shifts], "gt_depth": depth[i]})
        random.shuffle(samples)
        self.samples = samples[:int(percentage * len(samples))]

    def __getitem__(self, index):
        s = self.samples[index]
        imgs = [load_as_float(s["tgt"])] + ([load_as_float(r) for r in s["ref_imgs"]] if self.with_refs else [])
        gt = np.load(s["gt_depth"]).astype(np.float32)
        intr = np.copy(s["intrinsics"])
        if self.transform is not None:
            imgs, gt, intr = self.transform(imgs, gt, intr)
        if self.with_refs:
            return imgs[0], imgs[1:], intr, np.linalg.inv(intr), gt
        return imgs[0], gt

    def __len__(self):
        return len(self.samples)


class ValidationSet(data.Dataset):
    """root/scene/{0000000.jpg, 0000000.npy, ...} listed by val.txt -> (img, depth)."""

    def __init__(self, root, transform=None):
        self.scenes = _scenes(root, "val.txt")
        self.imgs, self.depth = [], []
        for scene in self.scenes:
            for img in _files(scene, ".jpg"):
                d = img[:-4] + ".npy"
                if not os.path.isfile(d):
                    raise FileNotFoundError("depth file {} not found".format(d))
                self.imgs.append(img)
                self.depth.append(d)
        self.transform = transform

    def __getitem__(self, index):
        img = load_as_float(self.imgs[index])
        depth = np.load(self.depth[index]).astype(np.float32)
        if self.transform is not None:
            t, _, _ = self.transform([img], depth, None)
            img = t[0]
        return img, torch.from_numpy(depth)

    def __len__(self):
        return len(self.imgs)


class SyntheticDepthSet(data.Dataset):
    """SURVEY.md section 8d inputs: image U(0,1) normalised with mean = std = 0.5; ground truth U(1, max) where a
    Bernoulli(density) mask is set, 0 elsewhere.  Deterministic per index."""

    def __init__(self, length, height=128, width=416, density=0.05, max_depth=80.0, seed=0, sequence_length=3, with_refs=False):
        self.length, self.h, self.w = int(length), height, width
        self.density, self.max_depth, self.seed = density, max_depth, seed
        self.nref = sequence_length - 1
        self.with_refs = with_refs
        self.scenes = ["synthetic"]

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 1000003 + index)
        img = (torch.rand(3, self.h, self.w, generator=g) - 0.5) / 0.5
        depth = torch.rand(self.h, self.w, generator=g) * (self.max_depth - 1.0) + 1.0
        gt = depth * (torch.rand(self.h, self.w, generator=g) < self.density).float()
        if not self.with_refs:
            return img, gt
        refs = [(img + 0.05 * torch.randn(3, self.h, self.w, generator=g)).clamp(-1, 1) for _ in range(self.nref)]
        intr = np.array([[241.67, 0, 204.17], [0, 246.28, 59.0], [0, 0, 1]], dtype=np.float32)
        return img, refs, intr, np.linalg.inv(intr), gt

    def __len__(self):
        return self.length


class RankSampler(data.Sampler):
    """Every rank draws the SAME shuffled global order and keeps its contiguous slice of each global batch -- the split
    nn.DataParallel.scatter makes of the reference's batch (train.py:316), one process per GPU instead of one per node.
    With drop_last=False (validation) the last, partial global batch is split into ceil(m / world)-sized contiguous slices
    (again what scatter does); a rank whose slice is empty simply has one batch fewer -- the validation loops carry no
    collective per batch, the (sum, count) all-reduce at their end handles unequal counts."""

    def __init__(self, n, global_batch, rank, world, shuffle, seed=0, drop_last=True):
        if global_batch % world != 0:
            raise ValueError("batch size %d does not divide over %d ranks" % (global_batch, world))
        self.n, self.gb, self.rank, self.world, self.shuffle, self.seed = n, global_batch, rank, world, shuffle, seed
        self.epoch = 0
        self.nbatches = n // global_batch if drop_last else (n + global_batch - 1) // global_batch

    def set_epoch(self, epoch):
        self.epoch = epoch

    def _slice(self, m):
        """[lo, hi) of this rank inside a global batch of m <= gb samples."""
        per = self.gb // self.world if m == self.gb else (m + self.world - 1) // self.world
        return min(self.rank * per, m), min((self.rank + 1) * per, m)

    def __iter__(self):
        order = list(range(self.n))
        if self.shuffle:
            random.Random(self.seed + self.epoch).shuffle(order)
        for b in range(self.nbatches):
            chunk = order[b * self.gb:(b + 1) * self.gb]
            lo, hi = self._slice(len(chunk))
            if hi > lo:
                yield chunk[lo:hi]

    def __len__(self):
        full = self.n // self.gb
        tail = self.n - full * self.gb
        if self.nbatches == full or tail == 0:
            return self.nbatches
        lo, hi = self._slice(tail)
        return full + (1 if hi > lo else 0)
