"""SURVEY 8 f-3 on the device: `dn_u8_normalize_flip` / `dn_flip_w` are BIT-EXACT against the reference's host transform chain
(custom_transforms.py: RandomHorizontalFlip :56-72, ArrayToTensor :40-53 incl. /255, Normalize :25-37), and ShardLoader batches equal
what data.Transform (the restatement of that chain used by the JPEG loader) yields for the same samples and flip draws."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import ctypes as C  # noqa: E402

from supervised_dispnet_amd import _lib, data as D, shards as S  # noqa: E402
from tests.cases import make_scene_folders  # noqa: E402

DEV = torch.device("cuda:0")


def _host_chain(u8_hwc, gt, flip, mean, std):
    """custom_transforms.py in its own order: fliplr copy -> transpose -> float()/255 -> sub_(m).div_(s)."""
    im = u8_hwc.astype(np.float32)
    if flip:
        im, gt = np.copy(np.fliplr(im)), np.copy(np.fliplr(gt))
    t = torch.from_numpy(np.transpose(im, (2, 0, 1)).copy()).float() / 255
    for ch, m, s in zip(t, mean, std):
        ch.sub_(m).div_(s)
    return t, torch.from_numpy(gt.copy()).float()


@pytest.mark.parametrize("b,h,w,imagenet", [(5, 16, 32, False), (3, 9, 13, True), (32, 128, 416, False), (2, 480, 640, True)])
def test_u8_normalize_flip_is_bit_exact(b, h, w, imagenet):
    r = np.random.RandomState(b * 1000 + w)
    u8 = r.randint(0, 256, (b, h, w, 3)).astype(np.uint8)
    gt = (r.rand(b, h, w) * 80).astype(np.float32)
    flips = (r.rand(b) < 0.5).astype(np.uint8)
    flips[0], flips[-1] = 1, 0
    mean, std = D.normalization(imagenet=imagenet)
    d_u8, d_fl, d_gt = torch.from_numpy(u8).to(DEV), torch.from_numpy(flips).to(DEV), torch.from_numpy(gt).to(DEV)
    out = torch.empty((b, 3, h, w), dtype=torch.float32, device=DEV)
    ogt = torch.empty_like(d_gt)
    md, sd = torch.tensor(mean, device=DEV), torch.tensor(std, device=DEV)
    mh, sh = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    st = torch.cuda.current_stream().cuda_stream
    _lib.call("dn_u8_normalize_flip", d_u8.data_ptr(), d_fl.data_ptr(), b, h, w, 3, md.data_ptr(), sd.data_ptr(), mh, sh, out.data_ptr(),
              3 * h * w, h * w, st)
    _lib.call("dn_flip_w", d_gt.data_ptr(), d_fl.data_ptr(), b, h, w, ogt.data_ptr(), st)
    torch.cuda.synchronize()
    for n in range(b):
        want, wgt = _host_chain(u8[n], gt[n], bool(flips[n]), mean, std)
        assert torch.equal(out[n].cpu(), want), "sample %d (flip %d)" % (n, flips[n])
        assert torch.equal(ogt[n].cpu(), wgt)
    # no flip vector at all == all zeros
    _lib.call("dn_u8_normalize_flip", d_u8.data_ptr(), None, b, h, w, 3, md.data_ptr(), sd.data_ptr(), mh, sh, out.data_ptr(), 3 * h * w, h * w, st)
    assert torch.equal(out[0].cpu(), _host_chain(u8[0], gt[0], False, mean, std)[0])


@pytest.mark.parametrize("with_refs", [False, True])
def test_shard_loader_equals_the_jpeg_loader_chain(tmp_path, with_refs):
    root = make_scene_folders(tmp_path / "kitti", frames=6, h=16, w=32)
    S.write_shards(str(root), str(tmp_path / "sh"), train=True, sequence_length=3)
    st = S.ShardSet(str(tmp_path / "sh"))
    mean, std = D.normalization()
    loader = S.ShardLoader(st, batch_size=5, device=DEV, mean=mean, std=std, flip=True, shuffle=False, with_refs=with_refs, drop_last=False,
                           flip_rng=random.Random(11))
    draws = random.Random(11)
    seen = 0
    for batch in loader:
        if with_refs:
            img, refs, k, kinv, gt = batch
        else:
            img, gt = batch
        for j in range(img.shape[0]):
            tgt, ref_idx, scene = st.samples[seen]
            flip = draws.random() < 0.5
            want, wgt = _host_chain(np.asarray(st.frames[tgt]), np.asarray(st.depth[tgt]), flip, mean, std)
            assert torch.equal(img[j].cpu(), want) and torch.equal(gt[j].cpu(), wgt)
            if with_refs:
                for r, fi in enumerate(ref_idx):
                    assert torch.equal(refs[r][j].cpu(), _host_chain(np.asarray(st.frames[fi]), np.asarray(st.depth[fi]), flip, mean, std)[0])
                kk = st.intrinsics[scene].copy()
                if flip:
                    kk[0, 2] = 32 - kk[0, 2]
                assert np.array_equal(k[j].cpu().numpy(), kk)
                np.testing.assert_allclose((k[j] @ kinv[j]).cpu().numpy(), np.eye(3), atol=1e-5)
            seen += 1
    assert seen == len(st) == 12 and len(loader) == 3
    # two ranks: the same global order, contiguous halves of every global batch of 4
    a = [b[0] for b in S.ShardLoader(st, 2, DEV, flip=False, shuffle=True, seed=3, rank=0, world=2)]
    b2 = [b[0] for b in S.ShardLoader(st, 2, DEV, flip=False, shuffle=True, seed=3, rank=1, world=2)]
    whole = [b[0] for b in S.ShardLoader(st, 4, DEV, flip=False, shuffle=True, seed=3)]
    assert len(a) == len(b2) == len(whole) == 3
    for x, y, z in zip(a, b2, whole):
        assert torch.equal(torch.cat((x, y)), z)
