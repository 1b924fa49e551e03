"""Closed-form inputs of the config-size parity cases shared by the CPU oracle tests and the GPU tests (the golden generator
tests/golden/make_goldens.py builds the same tensors from the same detgen tags)."""
import torch

from oracle import detgen


def config3_inputs(b=2, h=128, w=416):
    """BASELINE configs[2] (photometric warp loss, seq-len 3): target frame, two reference frames, KITTI intrinsics at 416x128."""
    tgt = detgen.image_batch(b, h, w, "cfg3:tgt")
    refs = [(tgt + 0.1 * detgen.uniform((b, 3, h, w), "cfg3:ref%d" % i, -1, 1)).clamp(-1, 1) for i in range(2)]
    k = torch.tensor([[241.67, 0, 204.17], [0, 246.28, 59.0], [0, 0, 1]], dtype=torch.float32).repeat(b, 1, 1)
    return tgt, refs, k, torch.inverse(k)


def dorn80_inputs(b=2, h=128, w=416):
    """BASELINE configs[4] (ordinal_c = 80): image, 5 %-dense ground truth, injected Dropout2d keep/scale pattern [b,16]."""
    x = detgen.image_batch(b, h, w, "dorn80:x")
    gt = detgen.sparse_depth(b, h, w, "dorn80:gt", density=0.05)
    mask = detgen.bernoulli((b, 16), "dorn80:drop", 0.5).float() * 2.0
    return x, gt, mask


# the rest of the DispNet zoo.  tag (tests/golden/zoo.npz prefix), product class, constructor kwargs, dataset of the loss,
# oracle call (oracle/nets_zoo.py)
def zoo_cases():
    from oracle import nets_zoo as Z
    return [
        ("res18", "Disp_res_18", {"datasets": "nyu"}, "nyu", lambda sd, x, tr: Z.disp_res_18(sd, x, training=tr, datasets="nyu")),
        ("res6", "Disp_res", {"datasets": "kitti"}, "kitti", lambda sd, x, tr: Z.disp_res6(sd, x, training=tr, datasets="kitti")),
        ("res101", "Disp_res_101", {"datasets": "kitti"}, "kitti",
         lambda sd, x, tr: Z.disp_res6(sd, x, training=tr, datasets="kitti", layer3_blocks=23, leaky=False)),
        ("vgg", "Disp_vgg", {"alpha": 10, "beta": 0.01}, "kitti", lambda sd, x, tr: Z.disp_vgg(sd, x, training=tr, alpha=10, beta=0.01)),
        ("vggfeat", "Disp_vgg_feature", {"datasets": "nyu", "with_classifier": False}, "nyu",
         lambda sd, x, tr: Z.disp_vgg(sd, x, training=tr, alpha=10, beta=0.1, layout="Disp_vgg_feature")),
    ]


# SURVEY 8 f-4 tail: FCRN's up-projection net and the dilated ASPP nets (tests/golden/zoo2.npz).  tag, product class, constructor kwargs,
# oracle call (training forward takes the injected Dropout2d pattern for FCRN)
def zoo2_cases():
    from oracle import nets_zoo as Z
    return [
        ("fcrn", "FCRN", {"datasets": "kitti"}, lambda sd, x, tr, mask=None: Z.fcrn(sd, x, training=tr, datasets="kitti", dropout_mask=mask)),
        ("res50_aspp", "res50_aspp", {"datasets": "kitti"}, lambda sd, x, tr, mask=None: Z.aspp_depth(sd, x, training=tr, counts=(3, 4, 6, 3))),
        ("deeplab", "deeplab_depth", {}, lambda sd, x, tr, mask=None: Z.aspp_depth(sd, x, training=tr, counts=(3, 4, 23, 3))),
    ]


def zoo2_dropout_mask(tag, b=2):
    return detgen.bernoulli((b, 64), "zoo2:fcrn:drop", 0.5).float() * 2.0 if tag == "fcrn" else None


def make_scene_folders(root, scenes=("s1", "s2", "s3"), frames=5, h=16, w=32, seed=0):
    """A tiny dataset in the reference's KITTI layout (scene/{%07d.jpg, %07d.npy, cam.txt} + train.txt / val.txt) with random content."""
    import numpy as np
    from PIL import Image
    r = np.random.RandomState(seed)
    for si, scene in enumerate(scenes):
        d = root / scene
        d.mkdir(parents=True)
        np.savetxt(d / "cam.txt", np.array([[100.0 + si, 0, w / 2.0 - 1.5], [0, 101.0, h / 2.0], [0, 0, 1]]))
        for i in range(frames):
            Image.fromarray(r.randint(0, 256, (h, w, 3)).astype(np.uint8)).save(d / ("%07d.jpg" % i), quality=95)
            np.save(d / ("%07d.npy" % i), (r.rand(h, w) * 80 * (r.rand(h, w) < 0.3)).astype(np.float32))
    (root / "train.txt").write_text("".join(s + "\n" for s in scenes))
    (root / "val.txt").write_text(scenes[-1] + "\n")
    return root
