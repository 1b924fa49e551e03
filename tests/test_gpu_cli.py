"""End-to-end run of the drop-in command lines on the GPU: train.py on synthetic samples (loss decreases, checkpoint with
the reference's keys), then test_disp.py's per-image evaluation on that checkpoint."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def test_train_then_evaluate_synthetic(tmp_path):
    import train
    import test_disp
    train.main(["SYN", "--synthetic", "16", "-b", "4", "--network", "disp_vgg_BN", "--loss", "L1", "--with-gt", "--epochs", "2",
                "--img-height", "64", "--img-width", "96", "--lr", "1e-3", "--save-root", str(tmp_path), "--print-freq", "100"])
    runs = [os.path.join(dp, f) for dp, _, fs in os.walk(tmp_path) for f in fs]
    ckpt = [r for r in runs if r.endswith("dispnet_checkpoint.pth.tar")]
    assert len(ckpt) == 1
    sd = torch.load(ckpt[0], map_location="cpu")
    assert set(sd.keys()) == {"epoch", "state_dict", "optimizer"} and sd["epoch"] == 2
    assert "features.features.0.weight" in sd["state_dict"] and "upconv4.0.weight" in sd["state_dict"]
    full = [r for r in runs if r.endswith("progress_log_full.csv")][0]
    rows = [l.split("\t") for l in open(full).read().strip().splitlines()]
    assert rows[0] == ["train_loss", "photo_loss", "explainability_loss", "smooth_loss"] and len(rows) == 1 + 2 * 4
    losses = [float(r[0]) for r in rows[1:]]
    assert all(np.isfinite(losses)) and np.mean(losses[4:]) < np.mean(losses[:4])
    summary = [r for r in runs if r.endswith("progress_log_summary.csv")][0]
    assert len(open(summary).read().strip().splitlines()) == 3

    # evaluation of one synthetic "KITTI" sample through test_disp.evaluate_sample
    import supervised_dispnet_amd.models as models
    import supervised_dispnet_amd.networks as networks
    import supervised_dispnet_amd.utils as U
    from supervised_dispnet_amd import kitti_eval as KE
    args = test_disp.build_parser().parse_args(["--network", "disp_vgg_BN", "--pretrained-dispnet", ckpt[0], "--img-height", "64",
                                                "--img-width", "96"])
    dev = torch.device("cuda")
    net = test_disp.create_disp_net(args, models, networks, dev)
    net.load_state_dict(sd["state_dict"])
    net.eval()
    r = np.random.RandomState(0)
    gt = np.where(r.rand(120, 180) < 0.2, r.uniform(1, 80, (120, 180)), 0.0)
    sample = {"tgt": r.uniform(0, 255, (120, 180, 3)).astype(np.float32), "gt_depth": gt, "mask": KE.generate_mask(gt, 1e-3, 80)}
    with torch.no_grad():
        errs, pred = test_disp.evaluate_sample(args, net, sample, dev, 1e-3, 80, KE, U)
    assert pred.shape == (64, 96) and len(errs) == 7 and all(np.isfinite(errs)) and 0 <= errs[4] <= errs[5] <= errs[6] <= 1


def test_evaluation_chain_matches_reference_golden(golden, tmp_path):
    """SURVEY 8 f-2 through the HIP network: test_disp.evaluate_sample (resize check -> normalise -> Disp_vgg_BN.eval() on the GPU -> 1/disp
    -> zoom -> clip -> Garg-crop mask -> scale factor -> 7 errors) equals the numbers the reference's own test_disp.py statements gave on
    the same sample and weights (tests/golden/eval_chain.npz), for the supervised, median-scaled and stereo branches."""
    import test_disp
    import supervised_dispnet_amd.models as models
    import supervised_dispnet_amd.utils as U
    from cases import check_eval_chain, eval_chain_sample
    from oracle import detgen
    from supervised_dispnet_amd import kitti_eval as KE
    g = golden("eval_chain")
    sample = eval_chain_sample(tmp_path)
    dev = torch.device("cuda")
    net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
    detgen.fill_state_dict(net.state_dict(), "vggbn")
    net.to(dev).eval()

    def evaluate(flags):
        args = test_disp.build_parser().parse_args(["--network", "disp_vgg_BN", "--pretrained-dispnet", "CKPT"] + flags)
        with torch.no_grad():
            return test_disp.evaluate_sample(args, net, sample, dev, 1e-3, 80, KE, U)

    check_eval_chain(g, evaluate, rtol=1e-3)


def _run_train(tmp_path, extra, epochs=2, n=8, b=4):
    import train
    train.main(["SYN", "--synthetic", str(n), "-b", str(b), "--epochs", str(epochs), "--img-height", "64", "--img-width", "96",
                "--lr", "1e-3", "--save-root", str(tmp_path), "--print-freq", "100"] + extra)
    runs = [os.path.join(dp, f) for dp, _, fs in os.walk(tmp_path) for f in fs]
    full = [r for r in runs if r.endswith("progress_log_full.csv")][0]
    rows = [l.split("\t") for l in open(full).read().strip().splitlines()]
    assert len(rows) == 1 + epochs * (n // b)
    vals = np.array([[float(v) for v in r] for r in rows[1:]])
    assert np.isfinite(vals).all()
    ckpt = [r for r in runs if r.endswith("dispnet_checkpoint.pth.tar")]
    assert len(ckpt) == 1
    return vals, torch.load(ckpt[0], map_location="cpu"), runs


def test_train_cli_multi_l1_default_loss(tmp_path):
    """--loss Multi_L1 is the reference's DEFAULT (train.py:34): whole-batch mask per scale, weights 1/2^i, bilinear GT pyramid."""
    vals, sd, _ = _run_train(tmp_path, ["--network", "disp_vgg_BN", "--with-gt"], epochs=3)       # no --loss: the default
    assert vals[:, 0].min() > 0 and np.mean(vals[-2:, 0]) < np.mean(vals[:2, 0])
    assert np.allclose(vals[:, 0], vals[:, 1])                                                   # -p 1 -m 0 -s 0: total = loss_1


def test_train_cli_dorn_loss(tmp_path):
    """--network disp_vgg_BN_DORN --loss DORN (train.py:431-433,466-468): SID labels, ordinal head, ordinal loss, and the
    validation's get_depth_sid decode (train.py:669-671)."""
    vals, sd, _ = _run_train(tmp_path, ["--network", "disp_vgg_BN_DORN", "--loss", "DORN", "--ordinal-c", "16", "--with-gt", "--seed", "3"],
                             epochs=3, n=16)
    assert "conv_ord.weight" in sd["state_dict"] and tuple(sd["state_dict"]["conv_ord.weight"].shape) == (32, 16, 1, 1)
    assert "disp0.0.weight" not in sd["state_dict"]
    # Dropout2d(0.5) on the 16 head channels draws a fresh mask every step (seeded: --seed pins the draws and the sample order, the
    # loader returns batches in sampler order whatever its workers do), so single steps are noisy: 12 steps, bounded AND directional
    # (a loss that only rises must fail).  The gradients themselves are pinned by tests/test_gpu_ordhead.py and the config-5 golden test
    assert np.isfinite(vals[:, 0]).all() and vals[:, 0].min() > 0 and vals[:, 0].max() < 1.5 * vals[0, 0]
    assert np.mean(vals[-4:, 0]) <= 1.02 * np.mean(vals[:2, 0]), vals[:, 0]


def test_train_cli_unsupervised_with_pose_training(tmp_path):
    """--unsupervised --train-pose: the 5-tuple loader the reference's branch needs (train.py:418-430), PoseExpNet -> photometric
    warp loss + explainability + smoothness (train.py:473-488), both nets in the optimizer, validate_without_gt."""
    vals, sd, runs = _run_train(tmp_path, ["--network", "disp_vgg_BN", "--unsupervised", "--train-pose", "-m", "0.2", "-s", "0.1",
                                           "--sequence-length", "3"], epochs=2)
    total, photo, expl, smooth = vals.T
    assert (photo > 0).all() and (expl > 0).all() and (smooth > 0).all()
    assert np.allclose(total, photo + 0.2 * expl + 0.1 * smooth, rtol=1e-5)
    pose_ckpt = [r for r in runs if r.endswith("exp_pose_checkpoint.pth.tar")]
    assert len(pose_ckpt) == 1
    psd = torch.load(pose_ckpt[0], map_location="cpu")["state_dict"]
    assert "pose_pred.weight" in psd and "predict_mask1.weight" in psd                           # -m > 0 builds the mask decoder
    summary = [r for r in runs if r.endswith("progress_log_summary.csv")][0]
    assert len(open(summary).read().strip().splitlines()) == 3


def test_train_cli_legacy_align_corners_changes_only_the_warp_sampling(tmp_path):
    """--legacy-align-corners threads align_corners=True through photometric_reconstruction_loss -> the warp kernel (reference
    inverse_warp.py:160-193 under torch 1.0.1): same data, same seed, the FIRST step's photometric loss equals the loss function called
    directly with that kwarg and differs from the default sampling's; everything else (smoothness term) is untouched."""
    from supervised_dispnet_amd import engine
    common = ["--network", "disp_vgg_BN", "--unsupervised", "-s", "0.1", "--sequence-length", "3"]
    v0, _, _ = _run_train(tmp_path / "a", common, epochs=1)
    v1, _, _ = _run_train(tmp_path / "b", common + ["--legacy-align-corners"], epochs=1)
    assert v0.shape == v1.shape and np.isfinite(v1).all()
    # The synthetic frames are the target + 5 % noise and the untrained pose net predicts ~zero motion, so the warp is (almost) the
    # identity: with align_corners=True the reference's pixel grid 2 X / (w - 1) - 1 (inverse_warp.py:62-68) lands exactly ON the
    # source pixels and the photometric term is the noise level; with today's default (False) every sample falls between pixels of a
    # noise image and the term is about twice that.  So the flag must LOWER the first step's photometric loss, by a lot.
    assert v1[0, 1] < 0.8 * v0[0, 1], (v0[0, 1], v1[0, 1])
    np.testing.assert_allclose(v0[0, 3], v1[0, 3], rtol=1e-6)                 # first step: same net, same smoothness value
    assert engine.compute_mode() == "f32x3"


def test_train_cli_compute_flag_selects_the_arithmetic(tmp_path):
    """--compute bf16 = BASELINE configs[4]'s mixed precision from the command line (not an env var): the run reports the mode, trains,
    and its first loss agrees with the fp32 default to the mode's stated tolerance; the library default is restored for later tests."""
    from supervised_dispnet_amd import engine
    common = ["--network", "disp_vgg_BN_DORN", "--loss", "DORN", "--ordinal-c", "16", "--with-gt"]
    try:
        v0, _, _ = _run_train(tmp_path / "a", common, epochs=1)
        assert engine.compute_mode() == "f32x3"
        v1, _, _ = _run_train(tmp_path / "b", common + ["--compute", "bf16"], epochs=1)
        assert engine.compute_mode() == "bf16"
        v2, _, _ = _run_train(tmp_path / "c", common + ["--compute", "f32"], epochs=1)
        assert engine.compute_mode() == "f32"
    finally:
        engine.set_compute("f32x3")
    # (Dropout2d draws differ between runs only through the seed, which is the same: the first losses are the same forward pass)
    np.testing.assert_allclose(v1[0, 0], v0[0, 0], rtol=5e-3)
    np.testing.assert_allclose(v2[0, 0], v0[0, 0], rtol=1e-4)


@pytest.mark.parametrize("network", ["disp_res_18", "disp_vgg", "disp_res_101"])
def test_train_cli_zoo_networks(tmp_path, network):
    """SURVEY 8 f-4 through the command line: --network disp_res_18 / disp_vgg (= models.Disp_vgg_feature, train.py:248) /
    disp_res_101 train, checkpoint with the reference's keys."""
    vals, sd, _ = _run_train(tmp_path, ["--network", network, "--loss", "L1", "--with-gt"], epochs=3)
    assert vals[:, 0].min() > 0 and np.mean(vals[-2:, 0]) < np.mean(vals[:2, 0])
    key = {"disp_res_18": "layer4.1.conv2.weight", "disp_vgg": "features.features.28.weight", "disp_res_101": "layer3.22.conv3.weight"}[network]
    assert key in sd["state_dict"]


@pytest.mark.parametrize("network", ["FCRN", "res50_aspp"])
def test_train_cli_fcrn_and_aspp_networks(tmp_path, network):
    """The tail of SURVEY 8 f-4 through the command line (train.py:251-256): --network FCRN (up-projection blocks, one output at the
    input size) and res50_aspp (dilated ResNet-50 + ASPP classifier, frozen BatchNorm affines) train and checkpoint with the
    reference's keys; the frozen BatchNorm pairs of the ASPP net stay at their initial values."""
    vals, sd, _ = _run_train(tmp_path, ["--network", network, "--loss", "L1", "--with-gt"], epochs=3)
    assert vals[:, 0].min() > 0
    key = {"FCRN": "up4.conv2_4.weight", "res50_aspp": "Scale.layer5.conv2d_list.3.weight"}[network]
    assert key in sd["state_dict"]
    if network == "res50_aspp":
        assert torch.all(sd["state_dict"]["Scale.layer3.0.bn2.weight"] == 1) and torch.all(sd["state_dict"]["Scale.bn1.bias"] == 0)
        assert not torch.all(sd["state_dict"]["Scale.layer3.0.bn2.running_mean"] == 0)       # ... while the statistics do move


def test_train_cli_from_uint8_shards(tmp_path):
    """SURVEY 8 f-3 through the command line: scene folders -> tools/make_shards.py -> train.py --shards (GPU-side flip / /255 / normalise);
    validation keeps reading the scene folders."""
    import train
    from supervised_dispnet_amd.shards import write_shards
    from tests.cases import make_scene_folders
    import pathlib
    root = make_scene_folders(pathlib.Path(tmp_path) / "kitti", frames=6, h=64, w=96)
    write_shards(str(root), str(tmp_path / "sh"), train=True, sequence_length=3)
    out = tmp_path / "run"
    train.main([str(root), "--shards", str(tmp_path / "sh"), "-b", "4", "--network", "disp_vgg_BN", "--loss", "L1", "--with-gt", "--epochs", "2",
                "--lr", "1e-3", "--save-root", str(out), "--print-freq", "100"])
    runs = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs]
    full = [r for r in runs if r.endswith("progress_log_full.csv")][0]
    rows = [l.split("\t") for l in open(full).read().strip().splitlines()]
    assert len(rows) == 1 + 2 * 3                                      # 12 samples / batch 4, two epochs
    vals = np.array([[float(v) for v in r] for r in rows[1:]])
    assert np.isfinite(vals).all() and (vals[:, 0] > 0).all()


@pytest.mark.parametrize("loss", ["L1", "Multi_L1"])
def test_train_cli_launch_tape_equals_the_eager_run(tmp_path, capsys, loss):
    """train.py (the launch tape is the DEFAULT for the supervised losses since round 6; --no-tape = eager launches): the supervised step
    through the launch tape (recorded on the first batch, checked bit for bit against the eager
    step on the second, replayed from the third on) logs the same per-step losses and ends with the same weights as the eager run of the
    same command line -- up to the last bit of Adam's step size (device-side vs host-side pow() of the bias corrections)."""
    args = ["--network", "disp_vgg_BN", "--with-gt", "--loss", loss, "--seed", "3"]
    ve, sde, _ = _run_train(tmp_path / "eager", args + ["--no-tape"], epochs=2, n=16, b=4)
    assert "--tape:" not in capsys.readouterr().out
    vt, sdt, _ = _run_train(tmp_path / "tape", args, epochs=2, n=16, b=4)                 # (no flag: the default)
    out = capsys.readouterr().out
    assert "--tape:" in out and "bit for bit" in out, out[-600:]
    assert "eager launches (" not in out
    assert vt.shape == ve.shape
    assert np.allclose(vt[:, 0], ve[:, 0], rtol=2e-5), (vt[:, 0], ve[:, 0])
    for k, v in sde["state_dict"].items():
        if torch.is_floating_point(v):
            assert torch.allclose(sdt["state_dict"][k], v, rtol=1e-4, atol=2e-6), k
