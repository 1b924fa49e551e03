"""Pre-decoded uint8 shards + the device-side transform chain.

The reference feeds the GPUs from `DataLoader(num_workers=batch_size)` workers that JPEG-decode with imageio, flip / transpose /
divide / normalise float32 images on the host (custom_transforms.py:25-72, datasets/sequence_folders.py:60-77, train.py:201-206) and
ship 12 B per pixel over PCIe.  At ~1 400 img/s per GPU that path starves the device.  Here

  * `write_shards` decodes a scene-folder dataset ONCE into flat files: `frames.u8` [n,H,W,3] (what imread returns), `depth.f32`
    [n,H,W] (the .npy ground truth, verbatim) and `meta.json` (samples in the reference's crawl order, per-scene intrinsics);
  * `ShardLoader` gathers a batch of frames from the memory-mapped files into pinned staging buffers on a prefetch thread, copies
    3 B per pixel to the device on a side stream, and ONE kernel (`dn_u8_normalize_flip`) produces the normalised fp32 NCHW batch with
    the per-sample RandomHorizontalFlip folded in (`dn_flip_w` mirrors the ground truth; the intrinsics' cx flips on the host).

The values are bit-identical to the host chain (`data.Transform`): ((float)u8 / 255 - mean) / std in IEEE fp32, same flip rule
(`rng.random() < 0.5`, one draw per sample).  The draws come from a PRIVATE `random.Random(seed + rank)` that `set_epoch` re-seeds
(seed, rank, epoch), so the flips are reproducible from --seed whatever else the process does with the global `random` module; a
caller that wants the JPEG loader's exact draws passes its own generator as `flip_rng` (tests do).
"""
import json
import os
import queue
import random
import threading

import numpy as np
import torch

from . import _lib, engine


def write_shards(root, out_dir, train=True, sequence_length=3, with_gt=False):
    """Decode `root` (scene folders listed by train.txt / val.txt, reference datasets/sequence_folders.py:14-58 resp.
    validation_folders.py:28-48 when `with_gt` and not `train`) into `out_dir`.  Frame order = crawl order; samples keep the crawl
    order too (the loader shuffles)."""
    from .data import _files, _scenes, load_as_float
    os.makedirs(out_dir, exist_ok=True)
    scenes = _scenes(root, "train.txt" if train else "val.txt")
    demi = (sequence_length - 1) // 2
    shifts = [s for s in range(-demi, demi + 1) if s != 0]
    frames, samples, intrinsics = [], [], []
    for si, scene in enumerate(scenes):
        cam = os.path.join(scene, "cam.txt")
        intr = np.genfromtxt(cam).astype(np.float32).reshape(3, 3) if os.path.isfile(cam) else np.eye(3, dtype=np.float32)
        intrinsics.append(intr.tolist())
        imgs = _files(scene, ".jpg")
        base = len(frames)
        for img in imgs:
            d = img[:-4] + ".npy"
            if not os.path.isfile(d):
                raise FileNotFoundError("depth file {} not found".format(d))
            frames.append((img, d))
        if with_gt and not train:
            samples += [[base + i, [], si] for i in range(len(imgs))]
        elif len(imgs) >= sequence_length:
            samples += [[base + i, [base + i + s for s in shifts], si] for i in range(demi, len(imgs) - demi)]
    if not frames:
        raise ValueError("no frames under {}".format(root))
    first = load_as_float(frames[0][0])
    H, W = first.shape[:2]
    fu8 = np.lib.format.open_memmap(os.path.join(out_dir, "frames.u8.npy"), mode="w+", dtype=np.uint8, shape=(len(frames), H, W, 3))
    fd = np.lib.format.open_memmap(os.path.join(out_dir, "depth.f32.npy"), mode="w+", dtype=np.float32, shape=(len(frames), H, W))
    for i, (img, d) in enumerate(frames):
        a = load_as_float(img)
        if a.shape != (H, W, 3):
            raise ValueError("{}: {} differs from the first frame's {}".format(img, a.shape, (H, W, 3)))
        fu8[i] = a.astype(np.uint8)                       # imread's own dtype: the float32 view holds exact 0..255 integers
        fd[i] = np.load(d).astype(np.float32)
    fu8.flush()
    fd.flush()
    meta = {"H": H, "W": W, "frames": len(frames), "samples": samples, "intrinsics": intrinsics, "sequence_length": sequence_length,
            "scenes": [os.path.basename(s) for s in scenes]}
    with open(os.path.join(out_dir, "meta.json"), "w") as f:
        json.dump(meta, f)
    return meta


class ShardSet(object):
    def __init__(self, shard_dir, percentage=1, seed=None):
        """`percentage` < 1 keeps that share of the samples after a seeded shuffle -- the reference's --data-amount
        (datasets/sequence_folders.py:40-41: random.shuffle(sequence_set); sequence_set[:int(percentage * len)])."""
        with open(os.path.join(shard_dir, "meta.json")) as f:
            self.meta = json.load(f)
        self.frames = np.load(os.path.join(shard_dir, "frames.u8.npy"), mmap_mode="r")
        self.depth = np.load(os.path.join(shard_dir, "depth.f32.npy"), mmap_mode="r")
        self.samples = self.meta["samples"]
        if percentage != 1:
            if not 0 < percentage <= 1:
                raise ValueError("--data-amount must lie in (0, 1], got %r" % (percentage,))
            order = list(self.samples)
            random.Random(seed).shuffle(order)
            self.samples = order[:int(percentage * len(order))]
            if not self.samples:
                raise ValueError("--data-amount %r keeps no sample of %d" % (percentage, len(order)))
        self.intrinsics = np.asarray(self.meta["intrinsics"], dtype=np.float32)
        self.H, self.W = self.meta["H"], self.meta["W"]
        self.scenes = self.meta.get("scenes", [])

    def __len__(self):
        return len(self.samples)


class ShardLoader(object):
    """Iterable over device-resident, normalised batches.  Yields (tgt_img[B,3,H,W], gt_depth[B,H,W]) or, `with_refs`, the 5-tuple
    (tgt_img, [ref_imgs], intrinsics[B,3,3], intrinsics_inv[B,3,3], gt_depth) of the reference's --unsupervised branch.
    One process per GPU: every rank walks the same shuffled order and keeps its contiguous slice of each global batch
    (data.RankSampler)."""

    def __init__(self, shards, batch_size, device, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), flip=True, shuffle=True, seed=0,
                 rank=0, world=1, with_refs=False, drop_last=True, prefetch=2, flip_rng=None, percentage=1):
        from .data import RankSampler
        self.set = shards if isinstance(shards, ShardSet) else ShardSet(shards, percentage=percentage, seed=seed)
        self.device = torch.device(device)
        engine.require_cuda(torch.empty(0, device=self.device), "ShardLoader device")
        self.B = batch_size
        self.flip, self.with_refs = flip, with_refs
        self.sampler = RankSampler(len(self.set), batch_size * world, rank, world, shuffle, seed=seed, drop_last=drop_last)
        self.mean_h = (C_float * 3)(*mean)
        self.std_h = (C_float * 3)(*std)
        self.mean_d = torch.tensor(mean, dtype=torch.float32, device=self.device)
        self.std_d = torch.tensor(std, dtype=torch.float32, device=self.device)
        self.prefetch = max(1, prefetch)
        self._own_rng = flip_rng is None
        self._seed, self._rank = int(seed or 0), int(rank)
        self.rng = flip_rng if flip_rng is not None else random.Random(self._flip_seed(0))
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.nimg = 1 + (len(self.set.samples[0][1]) if with_refs else 0)
        self._ring, self._slot = [], 0           # pinned staging buffers, reused round-robin once their copy has completed

    def _staging(self, b):
        n = self.prefetch + 2
        if not self._ring:
            H, W = self.set.H, self.set.W
            for _ in range(n):
                self._ring.append({"u8": torch.empty((self.nimg, self.B, H, W, 3), dtype=torch.uint8).pin_memory(),
                                   "gt": torch.empty((self.B, H, W), dtype=torch.float32).pin_memory(),
                                   "fl": torch.empty((self.B,), dtype=torch.uint8).pin_memory(), "ev": None})
        slot = self._ring[self._slot]
        self._slot = (self._slot + 1) % n
        if slot["ev"] is not None:
            slot["ev"].synchronize()             # the copy that last read this slot (prefetch + 2 batches ago) is done
        return slot

    def _flip_seed(self, epoch):
        return (self._seed * 1000003 + self._rank) * 1000003 + int(epoch)

    def set_epoch(self, epoch):
        self.sampler.set_epoch(epoch)
        if self._own_rng:
            self.rng = random.Random(self._flip_seed(epoch))

    def __len__(self):
        return len(self.sampler)

    # ---- producer thread: gather + host->device copies on the copy stream
    def _stage(self, idxs):
        s = self.set
        b, H, W = len(idxs), s.H, s.W
        slot = self._staging(b)
        u8, gt, fl = slot["u8"][:, :b], slot["gt"][:b], slot["fl"][:b]
        u8n, gtn, flips = u8.numpy(), gt.numpy(), fl.numpy()
        flips[:] = 0
        intr = np.empty((b, 3, 3), dtype=np.float32)
        for j, si in enumerate(idxs):
            tgt, refs, scene = s.samples[si]
            u8n[0, j] = s.frames[tgt]
            if self.with_refs:
                for r, fi in enumerate(refs):
                    u8n[1 + r, j] = s.frames[fi]
            gtn[j] = s.depth[tgt]
            k = s.intrinsics[scene].copy()
            if self.flip and self.rng.random() < 0.5:               # custom_transforms.py:61: one draw per sample
                flips[j] = 1
                k[0, 2] = W - k[0, 2]                               # :67-68
            intr[j] = k
        with torch.cuda.stream(self.copy_stream):
            d_u8 = torch.empty(u8.shape, dtype=torch.uint8, device=self.device)
            d_gt = torch.empty(gt.shape, dtype=torch.float32, device=self.device)
            d_fl = torch.empty(fl.shape, dtype=torch.uint8, device=self.device)
            if b == self.B:
                d_u8.copy_(u8, non_blocking=True)
            else:                                                    # ragged tail: the [:, :b] view of the ring buffer is strided
                for i in range(self.nimg):
                    d_u8[i].copy_(u8[i], non_blocking=True)
            d_gt.copy_(gt, non_blocking=True)
            d_fl.copy_(fl, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        slot["ev"] = ev
        return d_u8, d_gt, d_fl, intr, ev

    def _finish(self, staged):
        d_u8, d_gt, d_fl, intr, ev = staged
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        for t in (d_u8, d_gt, d_fl):
            t.record_stream(cur)
        n, b, H, W, _ = d_u8.shape
        imgs = torch.empty((n, b, 3, H, W), dtype=torch.float32, device=self.device)
        st = cur.cuda_stream
        for i in range(n):
            engine.hbm_call("dn::u8_norm_flip_vec_kernel", b * H * W * 15, "dn_u8_normalize_flip", d_u8[i].data_ptr(), d_fl.data_ptr(), b, H, W, 3,
                            self.mean_d.data_ptr(), self.std_d.data_ptr(), self.mean_h, self.std_h, imgs[i].data_ptr(), 3 * H * W, H * W, st)
        gt = torch.empty_like(d_gt)
        engine.hbm_call("dn::flip_w_kernel", b * H * W * 8, "dn_flip_w", d_gt.data_ptr(), d_fl.data_ptr(), b, H, W, gt.data_ptr(), st)
        if not self.with_refs:
            return imgs[0], gt
        k = torch.from_numpy(intr)
        return imgs[0], [imgs[i] for i in range(1, n)], k.to(self.device), torch.from_numpy(np.linalg.inv(intr)).to(self.device), gt

    def __iter__(self):
        q = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()

        def producer():
            try:
                torch.cuda.set_device(self.device)
                for idxs in self.sampler:
                    if stop.is_set():
                        return
                    q.put(self._stage(idxs))
                q.put(None)
            except BaseException as e:      # noqa: BLE001 -- surfaced on the consumer side
                q.put(e)

        t = threading.Thread(target=producer, daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield self._finish(item)
        finally:
            stop.set()
            while t.is_alive():
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                t.join(timeout=0.05)


import ctypes as _C  # noqa: E402

C_float = _C.c_float
