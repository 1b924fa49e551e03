"""Host-side logic of the drop-in command lines (train.py / test_disp.py) and the KITTI ground-truth generation:
flag surface of the reference (train.py:30-91, test_disp.py:23-49), the product's vectorised
duplicate-min scatter against the oracle's restatement of the reference loop and against the golden vectors, the
scene-folder readers.  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TRAIN_DEFAULTS = {
    "network": "disp_vgg", "dataset": "kitti", "imagenet_normalization": False, "pretrained_encoder": False, "loss": "Multi_L1",
    "ordinal_c": 80, "diff_lr": False, "sgd": False, "record": False, "unsupervised": False, "data_amount": 1, "monodepth2": False,
    "dataset_format": "sequential", "sequence_length": 3, "rotation_mode": "euler", "padding_mode": "zeros", "with_gt": False,
    "workers": 4, "epochs": 200, "epoch_size": 0, "batch_size": 4, "lr": 1e-4, "momentum": 0.9, "beta": 0.999, "weight_decay": 0,
    "print_freq": 10, "evaluate": False, "pretrained_disp": None, "pretrained_exp_pose": None, "seed": 0,
    "log_summary": "progress_log_summary.csv", "log_full": "progress_log_full.csv", "photo_loss_weight": 1, "mask_loss_weight": 0,
    "smooth_loss_weight": 0, "log_output": False, "training_output_freq": 0,
}
TEST_DEFAULTS = {
    "imagenet_normalization": False, "ordinal_c": 80, "unsupervised": False, "monodepth2": False, "pic": False, "error": False,
    "stereo": False, "mono": False, "pretrained_posenet": None, "img_height": 128, "img_width": 416, "no_resize": False,
    "dataset_dir": ".", "dataset_list": None, "output_dir": None, "gt_type": "KITTI", "img_exts": ["png", "jpg", "bmp"],
}


def test_train_cli_surface_matches_reference():
    import train
    p = train.build_parser()
    a = p.parse_args(["DATA"])
    for k, v in TRAIN_DEFAULTS.items():
        assert getattr(a, k) == v, k
    a = p.parse_args("DATA -b32 -m0.0 -s0.0 --loss L1 --network disp_vgg_BN --with-gt --wd 1e-5 --learning-rate 3e-4 -e "
                     "--pretrained-disp X --pretrained-exppose Y -p 2 -f 5 -j 9".split())
    assert (a.batch_size, a.loss, a.network, a.with_gt, a.weight_decay, a.lr, a.evaluate) == (32, "L1", "disp_vgg_BN", True, 1e-5, 3e-4, True)
    assert (a.pretrained_disp, a.pretrained_exp_pose, a.photo_loss_weight, a.training_output_freq, a.workers) == ("X", "Y", 2.0, 5, 9)
    assert set(train.NETWORKS) == {"dispnet", "disp_res", "disp_res_50", "disp_res_18", "disp_vgg", "disp_vgg_BN", "FCRN", "res50_aspp",
                                   "ASPP", "disp_res_101", "DORN", "disp_vgg_BN_DORN"}
    assert set(train.LOSSES) == {"Multi_L1", "Multi_full_L1", "Multi_berhu", "Multi_L2", "L1", "berhu", "L2", "scale_inv",
                                 "Multi_scale_inv", "DORN"}
    # README recipe -> folder name built from the non-default keys, in the reference's order (utils.py:11-43)
    name = train.save_path_formatter(a, p)
    assert name.split(os.sep)[0] == "DATA,b32,lr0.0003,p2.0,networkdisp_vgg_BN,lossL1"


def test_additive_flags_of_survey_section_5():
    """SURVEY.md section 5's additive flags: --legacy-align-corners (torch 1.0.1's grid_sample semantics at the reference's
    inverse_warp.py:191) and --compute (config 5's mixed precision from the command line); both default to the reference's behaviour."""
    import test_disp
    import train
    p = train.build_parser()
    a = p.parse_args(["DATA"])
    assert a.legacy_align_corners is False and a.compute is None
    a = p.parse_args(["DATA", "--legacy-align-corners", "--compute", "bf16"])
    assert a.legacy_align_corners is True and a.compute == "bf16"
    with pytest.raises(SystemExit):
        p.parse_args(["DATA", "--compute", "fp8"])
    # neither flag enters the run-folder name (utils.py:11-43 lists its keys explicitly)
    assert train.save_path_formatter(a, p).split(os.sep)[0] == "DATA"
    t = test_disp.build_parser().parse_args(["--network", "disp_vgg_BN", "--pretrained-dispnet", "CKPT", "--compute", "f32"])
    assert t.compute == "f32"


def test_models_refuse_multi_device_dataparallel_replication():
    """Reference train.py:316-317 wraps the net in nn.DataParallel.  On one device DataParallel calls the module itself (GPU test
    test_dataparallel_wrapper_on_one_device_equals_the_bare_module); beyond one device it replicates the module per forward, which the
    engine's per-process runtime state does not support: every drop-in model says so instead of silently sharing caches."""
    import supervised_dispnet_amd.models as models
    import supervised_dispnet_amd.networks as networks
    net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
    with pytest.raises(RuntimeError, match="one process per GPU"):
        net._replicate_for_data_parallel()
    for cls in (models.DispNetS, models.Disp_res_50, models.Disp_vgg_BN_DORN, models.PoseExpNet, models.FCRN, networks.DepthDecoder,
                networks.ResnetEncoder):
        assert cls._replicate_for_data_parallel is not torch.nn.Module._replicate_for_data_parallel, cls


def test_train_rejects_unknown_selectors():
    import train
    a = train.build_parser().parse_args(["DATA", "--loss", "nope"])
    with pytest.raises(ValueError):
        train.supervised_loss(a, None, None, None)
    a = train.build_parser().parse_args(["DATA", "--network", "nope"])
    with pytest.raises(ValueError):
        train.create_disp_net(a, None, None, "cpu")


def test_test_disp_cli_surface_matches_reference():
    import test_disp
    p = test_disp.build_parser()
    with pytest.raises(SystemExit):
        p.parse_args([])                                     # --network and --pretrained-dispnet are required
    a = p.parse_args(["--network", "disp_vgg_BN", "--pretrained-dispnet", "CKPT"])
    for k, v in TEST_DEFAULTS.items():
        assert getattr(a, k) == v, k


def test_checkpoint_files_and_keys(tmp_path):
    import train
    st = {"epoch": 1, "state_dict": {"w": torch.zeros(1)}, "optimizer": {"step": 1}}
    train.save_checkpoint(str(tmp_path), st, {"epoch": 1, "state_dict": {}}, is_best=True, epoch=0, record=True)
    names = sorted(os.listdir(tmp_path))
    assert names == ["dispnet_checkpoint.pth.tar", "dispnet_model_best.pth.tar", "exp_pose_checkpoint.pth.tar",
                     "exp_pose_model_best.pth.tar", "weights_0"]
    assert set(torch.load(tmp_path / "dispnet_model_best.pth.tar").keys()) == {"epoch", "state_dict", "optimizer"}


def _cloud(seed, n, h, w):
    r = np.random.RandomState(seed)
    # many points per pixel and integer-ish coordinates so duplicates AND (n-1)-linearisation collisions are common
    pts = np.stack([r.randint(0, w, n).astype(np.float64), r.randint(0, h, n).astype(np.float64), r.uniform(-2, 80, n)], 1)
    return pts


@pytest.mark.parametrize("seed,n,h,w", [(0, 5000, 24, 40), (1, 20000, 37, 124), (2, 50, 8, 9), (3, 0, 5, 7)])
def test_vectorised_duplicate_min_scatter_is_bit_exact(seed, n, h, w):
    from oracle import kitti_gt as OK
    from supervised_dispnet_amd import kitti_eval as KE
    pts = _cloud(seed, n, h, w)
    got = KE.scatter_depth_min_duplicates(pts, (h, w))
    want = OK.scatter_depth_min_duplicates(pts, (h, w))
    assert got.dtype == want.dtype and np.array_equal(got, want)


def test_kitti_ground_truth_matches_reference_golden(golden, tmp_path):
    """The product's file-reading path (calibration text, velodyne .bin) on the synthetic scene whose reference outputs are
    the committed golden vectors."""
    import importlib.util
    import pathlib
    from supervised_dispnet_amd import kitti_eval as KE
    g = golden("kitti_gt")
    spec = importlib.util.spec_from_file_location("mk", pathlib.Path(__file__).parent / "golden" / "make_goldens.py")
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    p_rect, r_rect, r, t, velo = mk.synthetic_kitti_scene()
    fmt = lambda a: " ".join("%.6e" % v for v in a)
    (tmp_path / "calib_cam_to_cam.txt").write_text("calib_time: 09-Jan-2012 13:57:47\nR_rect_00: %s\nP_rect_02: %s\n" % (fmt(r_rect), fmt(p_rect)))
    (tmp_path / "calib_velo_to_cam.txt").write_text("calib_time: 15-Mar-2012 11:37:16\nR: %s\nT: %s\n" % (fmt(r), fmt(t)))
    velo.astype(np.float32).tofile(tmp_path / "0000000000.bin")
    for shape in ((375, 1242), (120, 400)):
        depth = KE.generate_depth_map(str(tmp_path), str(tmp_path / "0000000000.bin"), shape, cam=2)
        mask = KE.generate_mask(depth, 1e-3, 80)
        yy, xx = np.nonzero(depth)
        np.testing.assert_array_equal(np.stack([yy, xx], 1).astype(np.int32), g["depth:%dx%d:yx" % shape])
        np.testing.assert_array_equal(depth[yy, xx], g["depth:%dx%d:val" % shape])            # bit-exact float64
        assert int(mask.sum()) == int(g["mask:%dx%d:count" % shape])
        np.testing.assert_array_equal(mask.sum(1).astype(np.int32), g["mask:%dx%d:rowsum" % shape])
    assert tuple(KE.garg_crop(375, 1242)) == (153, 371, 44, 1197)
    assert tuple(KE.garg_crop(128, 416)) == (52, 126, 14, 401)


def test_abs_rel_per_pixel_and_worst_300_match_reference_golden(golden, tmp_path):
    """test_disp.py:471-477 + :318-350 (--error): the per-pixel abs-rel map, the np.where / argpartition(-300) selection and the
    annotation, against vectors produced by executing the reference's own statements (tests/golden/make_goldens.py::gold_worst_pixels)
    and against the oracle's sort-based restatement.  Index work: exact."""
    import importlib.util
    import pathlib
    from oracle import kitti_gt as OK
    from supervised_dispnet_amd import kitti_eval as KE
    g = golden("worst_pixels")
    spec = importlib.util.spec_from_file_location("mk", pathlib.Path(__file__).parent / "golden" / "make_goldens.py")
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    p_rect, r_rect, r, t, velo = mk.synthetic_kitti_scene()
    fmt = lambda a: " ".join("%.6e" % v for v in a)
    (tmp_path / "calib_cam_to_cam.txt").write_text("calib_time: 09-Jan-2012 13:57:47\nR_rect_00: %s\nP_rect_02: %s\n" % (fmt(r_rect), fmt(p_rect)))
    (tmp_path / "calib_velo_to_cam.txt").write_text("calib_time: 15-Mar-2012 11:37:16\nR: %s\nT: %s\n" % (fmt(r), fmt(t)))
    velo.astype(np.float32).tofile(tmp_path / "0000000000.bin")
    shape = (375, 1242)
    gt = KE.generate_depth_map(str(tmp_path), str(tmp_path / "0000000000.bin"), shape, cam=2)
    for variant in (0, 1):
        tag = "%dx%d:v%d" % (shape + (variant,))
        pred = mk.abs_rel_inputs(shape, variant)
        m = KE.compute_abs_rel_per_pixel(gt, pred, 1e-3, 80)
        om = OK.compute_abs_rel_per_pixel(gt, pred, 1e-3, 80)
        np.testing.assert_array_equal(m, om)
        yy, xx = np.nonzero(m > 0)
        np.testing.assert_array_equal(np.stack([yy, xx], 1).astype(np.int32), g["abs_rel:%s:yx" % tag])
        np.testing.assert_array_equal(m[yy, xx], g["abs_rel:%s:val" % tag])                     # bit-exact float64
        assert int((m == -1).sum()) == int(g["abs_rel:%s:neg_count" % tag])
        gi, index_result = KE.worst_pixels(m, 300)
        assert gi.dtype == np.int32 and gi.shape == (300, 2)
        assert index_result.shape[0] == int(g["worst:%s:n_valid" % tag])
        np.testing.assert_array_equal(gi, g["worst:%s:graph_index" % tag])                      # same numpy routine, same order
        want_set, v300, v301 = OK.worst_pixels_loop(om, 300)
        assert v300 > v301, "the synthetic case must not tie at the selection boundary"
        assert sorted(map(tuple, gi.tolist())) == want_set                                       # order-free check vs the sort-based oracle
        tgt = (np.arange(shape[0] * shape[1] * 3, dtype=np.int64) * 7919 % 256).reshape(shape + (3,)).astype(np.float32)
        ann = KE.annotate_pixels(tgt, gi)
        np.testing.assert_array_equal(ann.astype(np.float64).sum(axis=(0, 1)), g["annotate:%s:sum" % tag])
        got = np.stack([ann[y:y + 5, x:x + 5] for y, x in zip(gi[:8, 0], gi[:8, 1])])
        np.testing.assert_array_equal(got, g["annotate:%s:patch" % tag])
    # the threshold branch (test_disp.py:339-342) and the too-few-pixels error
    gi03, _ = KE.worst_pixels(m, threshold=0.3)
    assert gi03.shape[0] == int((m[153:371, 44:1197] > 0.3).sum())
    with pytest.raises(ValueError):
        KE.worst_pixels(np.full((20, 30), -1.0), 300)


def test_evaluation_chain_matches_reference_golden_on_cpu(golden, tmp_path):
    """SURVEY 8 f-2, the chain between the network and "Abs Rel" (reference test_disp.py:178-398, inline in its main()):
    test_disp.evaluate_sample -- transpose, /255, normalise, forward, 1/disp, scipy zoom to the ground-truth size, clip, Garg-crop mask,
    scale factor (1 / median ratio / 5.4), compute_errors -- against numbers the reference's own statements produced on the same
    deterministic sample (tests/golden/make_goldens.py::gold_eval_chain).  Here the forward is the ORACLE's eval-mode Disp_vgg_BN on
    the CPU (the checker standing in for the network, so the host chain is pinned without a GPU); tests/test_gpu_cli.py runs the same
    check through the HIP network."""
    import test_disp
    import supervised_dispnet_amd.utils as U
    from cases import check_eval_chain, eval_chain_sample
    from oracle import detgen, nets as ON
    from supervised_dispnet_amd import kitti_eval as KE
    g = golden("eval_chain")
    sample = eval_chain_sample(tmp_path)
    assert int(sample["mask"].sum()) == int(g["n_valid"])
    sd = ON.disp_vgg_bn_state_dict()
    detgen.fill_state_dict(sd, "vggbn")

    class _OracleNet(object):
        def __call__(self, t):
            with torch.no_grad():
                return ON.disp_vgg_bn(sd, t, training=False)

    def evaluate(flags):
        args = test_disp.build_parser().parse_args(["--network", "disp_vgg_BN", "--pretrained-dispnet", "CKPT"] + flags)
        return test_disp.evaluate_sample(args, _OracleNet(), sample, torch.device("cpu"), 1e-3, 80, KE, U)

    check_eval_chain(g, evaluate, rtol=2e-5)                      # same fp32 CPU arithmetic as the reference's forward: near-exact


def test_scene_folder_readers_and_rank_sampler(tmp_path):
    from PIL import Image
    from supervised_dispnet_amd import data as D
    root = tmp_path / "kitti"
    for scene in ("s1", "s2"):
        d = root / scene
        d.mkdir(parents=True)
        np.savetxt(d / "cam.txt", np.array([[100.0, 0, 8], [0, 100.0, 4], [0, 0, 1]]))
        for i in range(4):
            Image.fromarray(np.full((8, 16, 3), 10 * i, dtype=np.uint8)).save(d / ("%07d.jpg" % i), quality=100)
            np.save(d / ("%07d.npy" % i), np.full((8, 16), float(i + 1), dtype=np.float32))
    (root / "train.txt").write_text("s1\ns2\n")
    (root / "val.txt").write_text("s2\n")
    mean, std = D.normalization()
    ds = D.SequenceFolder(str(root), seed=0, train=True, sequence_length=3, transform=D.Transform(mean, std, flip=True))
    assert len(ds) == 4                                               # 2 scenes x (4 - 2) centre frames
    img, gt = ds[0]
    assert img.shape == (3, 8, 16) and gt.shape == (8, 16) and img.dtype == torch.float32
    assert float(img.min()) >= -1.0 - 1e-6 and float(img.max()) <= 1.0 + 1e-6
    five = D.SequenceFolder(str(root), seed=0, train=True, sequence_length=3, transform=D.Transform(mean, std, flip=False), with_refs=True)[0]
    assert len(five) == 5 and len(five[1]) == 2 and np.allclose(five[2] @ five[3], np.eye(3), atol=1e-5)
    val = D.ValidationSet(str(root), transform=D.Transform(mean, std, flip=False))
    assert len(val) == 4 and val[1][1].shape == (8, 16)
    # global batch 4 over 2 ranks: same global order, contiguous halves
    a = list(D.RankSampler(8, 4, 0, 2, shuffle=True, seed=3))
    b = list(D.RankSampler(8, 4, 1, 2, shuffle=True, seed=3))
    assert len(a) == 2 and sorted(sum(a, []) + sum(b, [])) == list(range(8))
    whole = list(D.RankSampler(8, 4, 0, 1, shuffle=True, seed=3))
    assert [x + y for x, y in zip(a, b)] == whole
    s = D.SyntheticDepthSet(3)
    assert torch.equal(s[1][0], s[1][0]) and 0.02 < float((s[0][1] > 0).float().mean()) < 0.08


def test_shard_writer_matches_the_scene_folder_reader(tmp_path):
    """the pre-decoded uint8 shards hold exactly what the reference's loader decodes (imread frames, .npy depth, cam.txt)
    and the same sample list as datasets/sequence_folders.py:32-50 crawls (before its shuffle)."""
    from supervised_dispnet_amd import data as D
    from supervised_dispnet_amd import shards as S
    from tests.cases import make_scene_folders
    root = make_scene_folders(tmp_path / "kitti")
    meta = S.write_shards(str(root), str(tmp_path / "sh"), train=True, sequence_length=3)
    st = S.ShardSet(str(tmp_path / "sh"))
    assert meta["frames"] == 15 and len(st) == 9 and (st.H, st.W) == (16, 32)
    ds = D.SequenceFolder(str(root), seed=0, train=True, sequence_length=3, transform=None, with_refs=True)
    crawl = sorted((s["tgt"], tuple(s["ref_imgs"])) for s in ds.samples)
    names = []
    for scene in ("s1", "s2", "s3"):
        names += [str(root / scene / ("%07d.jpg" % i)) for i in range(5)]
    assert sorted((names[t], tuple(names[r] for r in refs)) for t, refs, _ in st.samples) == crawl
    for fi in (0, 7, 14):
        a = D.load_as_float(names[fi])
        assert np.array_equal(st.frames[fi].astype(np.float32), a)                 # imread's integers, losslessly as uint8
        assert np.array_equal(st.depth[fi], np.load(names[fi][:-4] + ".npy"))
    np.testing.assert_array_equal(st.intrinsics[1], np.genfromtxt(root / "s2" / "cam.txt").astype(np.float32))
    val = S.write_shards(str(root), str(tmp_path / "shv"), train=False, with_gt=True)
    assert len(val["samples"]) == 5 and val["samples"][2] == [2, [], 0]


def test_imresize_matches_scipy_misc_semantics():
    from supervised_dispnet_amd import kitti_eval as KE
    img = np.zeros((4, 4, 3), dtype=np.float32)
    img[:, 2:] = 255.0
    out = KE.imresize_bilinear(img, (2, 2))        # PIL's bilinear reduce (support scales with the factor), as scipy.misc used
    assert out.shape == (2, 2, 3) and out.dtype == np.uint8 and out[0, 0, 0] < 64 and out[0, 1, 0] > 192
    # float input is byte-scaled by its own min / max first (scipy.misc.toimage): [10, 20] -> [0, 255]
    ramp = np.repeat(np.linspace(10.0, 20.0, 4, dtype=np.float32)[None, :, None], 3, 2).repeat(4, 0)
    same = KE.imresize_bilinear(ramp, (4, 4))
    assert same[0, 0, 0] == 0 and same[0, 3, 0] == 255
