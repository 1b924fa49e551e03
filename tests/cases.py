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


def _golden_generator():
    """tests/golden/make_goldens.py as a module: only its closed-form input builders are used by the tests (nothing there that is
    called from here touches /root/reference)."""
    import importlib.util
    import pathlib
    spec = importlib.util.spec_from_file_location("mk", pathlib.Path(__file__).parent / "golden" / "make_goldens.py")
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    return mk


def eval_chain_sample(tmp_path):
    """The deterministic sample of tests/golden/eval_chain.npz (make_goldens.py::gold_eval_chain): closed-form image at the network's
    input size, ground truth + Garg-crop mask of the synthetic KITTI scene through the PRODUCT's kitti_eval (bit-exact against the
    reference's: test_kitti_ground_truth_matches_reference_golden)."""
    import numpy as np
    from supervised_dispnet_amd import kitti_eval as KE
    mk = _golden_generator()
    p_rect, r_rect, r, t, velo = mk.synthetic_kitti_scene()
    fmt = lambda a: " ".join("%.6e" % v for v in a)
    (tmp_path / "calib_cam_to_cam.txt").write_text("calib_time: 09-Jan-2012 13:57:47\nR_rect_00: %s\nP_rect_02: %s\n" % (fmt(r_rect), fmt(p_rect)))
    (tmp_path / "calib_velo_to_cam.txt").write_text("calib_time: 15-Mar-2012 11:37:16\nR: %s\nT: %s\n" % (fmt(r), fmt(t)))
    velo.astype(np.float32).tofile(tmp_path / "0000000000.bin")
    gt = KE.generate_depth_map(str(tmp_path), str(tmp_path / "0000000000.bin"), (375, 1242), cam=2)
    return {"tgt": mk.eval_chain_image(128, 416), "gt_depth": gt, "mask": KE.generate_mask(gt, 1e-3, 80)}


def check_eval_chain(g, evaluate, rtol=1e-3):
    """`evaluate(flag list) -> (7 errors, depth at network resolution)` against the reference's numbers for the three scale-factor
    branches of test_disp.py:391-396.  Tolerances: the seven errors are means over 8961 valid pixels of an fp32 network's output --
    rtol 1e-3 (the fp32 tolerance of the disparity maps); the three threshold accuracies count pixels, so a pixel within rounding of
    1.25^k may flip: |a_k - ref| <= 3 / n_valid on top."""
    import numpy as np
    n = int(g["n_valid"])
    for name, flags in (("supervised", []), ("median", ["--unsupervised"]), ("stereo", ["--stereo"])):
        errs, depth = evaluate(flags)
        want = g["errors:" + name]
        np.testing.assert_allclose(np.asarray(errs[:4], np.float64), want[:4], rtol=rtol, err_msg=name)
        assert np.all(np.abs(np.asarray(errs[4:], np.float64) - want[4:]) <= rtol * want[4:] + 3.0 / n), (name, errs[4:], want[4:])
        if name == "supervised":
            assert depth.shape == (128, 416)
            np.testing.assert_allclose(depth.reshape(-1)[::97], g["pred_depth_samples"], rtol=rtol)
            np.testing.assert_allclose(float(depth.astype(np.float64).sum()), float(g["pred_depth_sum"]), rtol=1e-4)
