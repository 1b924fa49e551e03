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
