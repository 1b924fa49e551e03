#!/usr/bin/env python3
"""can the input pipeline keep one MI355X fed at the training rate (~1 400 img/s, Disp_vgg_BN 128x416 b32)?

  loader_only      ShardLoader (uint8 shards -> pinned staging -> 3 B/px H2D -> dn_u8_normalize_flip) iterated flat out
  host_chain       the reference's chain (custom_transforms.py flip / transpose / /255 / normalise in float32 on DataLoader workers,
                   12 B/px H2D) over the SAME decoded frames -- JPEG decode excluded, i.e. an upper bound for the reference's loader
  train_resident   training step on one batch resident in HBM (what bench.py times)
  train_shards     the same training step fed by the ShardLoader

usage: python tools/loader_bench.py [--frames 2048] [--batch 32] [--steps 40] [--workers 16]      (one JSON line)"""
import argparse, json, os, pathlib, sys, tempfile, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=2048)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--workers", type=int, default=16)
a = ap.parse_args()

import bench
import supervised_dispnet_amd.loss_functions as LF
import supervised_dispnet_amd.models as models
from supervised_dispnet_amd import data as D, shards as S
from supervised_dispnet_amd.functional import reciprocal
from supervised_dispnet_amd.optim import FusedAdam

dev = torch.device("cuda:0")
H, W, B = 128, 416, a.batch
tmp = tempfile.mkdtemp(prefix="dn_shards_")
r = np.random.RandomState(0)
n = a.frames
fu8 = np.lib.format.open_memmap(os.path.join(tmp, "frames.u8.npy"), mode="w+", dtype=np.uint8, shape=(n, H, W, 3))
fd = np.lib.format.open_memmap(os.path.join(tmp, "depth.f32.npy"), mode="w+", dtype=np.float32, shape=(n, H, W))
for i in range(0, n, 64):
    fu8[i:i + 64] = r.randint(0, 256, (min(64, n - i), H, W, 3), dtype=np.uint8)
    d = r.rand(min(64, n - i), H, W).astype(np.float32)
    fd[i:i + 64] = np.where(r.rand(*d.shape) < 0.05, d * 79 + 1, 0).astype(np.float32)
json.dump({"H": H, "W": W, "frames": n, "samples": [[i, [], 0] for i in range(n)], "intrinsics": [np.eye(3).tolist()], "sequence_length": 1,
           "scenes": ["synthetic"]}, open(os.path.join(tmp, "meta.json"), "w"))
st = S.ShardSet(tmp)
out = {"frames": n, "batch": B, "host_logical_cpus": os.cpu_count()}


def rate(it, steps, per=B):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 0
    for _ in it:
        k += 1
        if k == steps:
            break
    torch.cuda.synchronize()
    return k * per / (time.perf_counter() - t0)


loader = S.ShardLoader(st, B, dev, flip=True, shuffle=True, seed=0)
rate(loader, 5)
out["loader_only_img_s"] = rate(loader, min(a.steps * 2, len(loader)))


class HostChain(torch.utils.data.Dataset):
    def __init__(self):
        self.t = D.Transform(*D.normalization(), flip=True)

    def __len__(self):
        return n

    def __getitem__(self, i):
        imgs, gt, _ = self.t([np.asarray(st.frames[i]).astype(np.float32)], np.asarray(st.depth[i]), np.eye(3, dtype=np.float32))
        return imgs[0], gt


def host_iter(workers):
    dl = torch.utils.data.DataLoader(HostChain(), batch_size=B, shuffle=True, num_workers=workers, pin_memory=True, drop_last=True)
    for img, gt in dl:
        yield img.to(dev, non_blocking=True), gt.to(dev, non_blocking=True)


out["host_chain_workers"] = a.workers
out["host_chain_img_s"] = rate(host_iter(a.workers), min(a.steps, n // B))

torch.manual_seed(0)
net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
bench._quiet_init(net)
net.to(dev).train()
opt = FusedAdam(net._hot_parameters(), lr=1e-4, production_order=net._grad_production_order())


def step(img, gt):
    depth = [reciprocal(d) for d in net(img)]
    loss = LF.l1_loss(gt, depth, "kitti")
    opt.zero_grad()
    loss.backward()
    opt.step()


img0, gt0 = bench.synthetic_batch(B, H, W, dev, 0)
for _ in range(5):
    step(img0, gt0)


def resident():
    while True:
        step(img0, gt0)
        yield None


def fed():
    while True:
        for img, gt in loader:
            step(img, gt)
            yield None


out["train_resident_img_s"] = rate(resident(), a.steps)
rate(fed(), 3)
out["train_shards_img_s"] = rate(fed(), a.steps)
out["train_shards_over_resident"] = out["train_shards_img_s"] / out["train_resident_img_s"]
print(json.dumps(out))
