#!/usr/bin/env python3
"""Depth evaluation on the KITTI Eigen split / NYU -- the command line of the reference's test_disp.py (:23-49), running the
network on the MI355X HIP path:

    python3 test_disp.py --pretrained-dispnet CKPT --network disp_vgg_BN --dataset-dir KITTI_RAW \
        --dataset-list kitti_eval/test_files_eigen.txt

Per image (test_disp.py:184-450): resize to --img-height x --img-width (scipy.misc.imresize semantics), /255 and normalise,
forward in eval mode, depth = 1/disp, cubic-spline zoom to the ground-truth size clipped to [min_depth, max_depth], mask
(valid range AND Garg crop), optional median scaling (--unsupervised / --mono) or x5.4 (--stereo), 7 error metrics.
--error: the per-pixel abs-rel map and the 300 worst pixels inside the Garg crop (test_disp.py:309-373: compute_abs_rel_per_pixel,
np.argpartition(..., -300)), annotated input written under output/[stereo|mono/]bad_300pixel/ like the reference.
The PoseNet-scaled evaluation (--pretrained-posenet) and the --pic comparison plots are outside this path.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def build_parser():
    p = argparse.ArgumentParser(description="Script for DispNet testing with corresponding groundTruth",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--network", required=True, type=str, help="network type")
    p.add_argument("--imagenet-normalization", action="store_true", help="use imagenet parameter for normalization.")
    p.add_argument("--ordinal-c", default=80, type=int, metavar="N", help="DORN loss channel number")
    p.add_argument("--unsupervised", action="store_true", help="to have unsupervised loss")
    p.add_argument("--monodepth2", action="store_true", help="to test direct finetuned monodepth2 model")
    p.add_argument("--pic", action="store_true", help="to store performance comparison pics")
    p.add_argument("--error", action="store_true", help="to store performance over different pics")
    p.add_argument("--stereo", action="store_true", help="to test monodepth2 stereo model")
    p.add_argument("--mono", action="store_true", help="to test monodepth2 mono video model")
    p.add_argument("--pretrained-dispnet", required=True, type=str, help="pretrained DispNet path")
    p.add_argument("--pretrained-posenet", default=None, type=str, help="pretrained PoseNet path (for scale factor)")
    p.add_argument("--img-height", default=128, type=int, help="Image height")
    p.add_argument("--img-width", default=416, type=int, help="Image width")
    p.add_argument("--no-resize", action="store_true", help="no resizing is done")
    p.add_argument("--dataset-dir", default=".", type=str, help="Dataset directory")
    p.add_argument("--dataset-list", default=None, type=str, help="Dataset list file")
    p.add_argument("--output-dir", default=None, type=str, help="Output directory for saving predictions in a big 3D numpy file")
    p.add_argument("--gt-type", default="KITTI", type=str, help="GroundTruth data type", choices=["npy", "png", "KITTI", "NYU", "stillbox"])
    p.add_argument("--img-exts", default=["png", "jpg", "bmp"], nargs="*", type=str, help="images extensions to glob")
    # extension (not in the reference)
    p.add_argument("--compute", choices=["f32x3", "f32", "bf16"], default=None,
                   help="arithmetic of the matrix-core convolution kernels (see train.py --compute); default: the library's f32x3")
    return p


def create_disp_net(args, models, networks, device, U=None):
    if args.monodepth2 or args.stereo or args.mono:
        if args.network == "disp_vgg_BN":
            enc = networks.vggEncoder(num_layers=16, pretrained=False).to(device)
        elif args.network == "disp_res_18":
            enc = networks.ResnetEncoder(num_layers=18, pretrained=False).to(device)
        else:
            raise ValueError("undefined network")
        dec = networks.DepthDecoder(enc.num_ch_enc).to(device)
        if args.mono or args.stereo:            # test_disp.py:85-86: the monodepth2 weights come from a folder (encoder.pth, depth.pth)
            U.load_model({"encoder": enc, "depth": dec}, args.pretrained_dispnet)
        return models.monodepth2(encoder=enc, decoder=dec)
    table = {"dispnet": "DispNetS", "disp_res": "Disp_res", "disp_res_50": "Disp_res_50", "disp_res_18": "Disp_res_18",
             "disp_vgg": "Disp_vgg_feature", "disp_vgg_BN": "Disp_vgg_BN", "FCRN": "FCRN", "res50_aspp": "res50_aspp",
             "ASPP": "deeplab_depth", "disp_res_101": "Disp_res_101", "DORN": "DORN"}
    if args.network in table:
        return getattr(models, table[args.network])().to(device)
    if args.network == "disp_vgg_BN_DORN":
        return models.Disp_vgg_BN_DORN(ordinal_c=args.ordinal_c).to(device)
    raise ValueError("undefined network")


def evaluate_sample(args, disp_net, sample, device, min_depth, max_depth, KE, U):
    """One image -> (7 errors, predicted depth at network resolution)."""
    from scipy.ndimage import zoom
    from supervised_dispnet_amd.data import normalization
    tgt = sample["tgt"]
    if args.gt_type == "NYU":
        tgt = np.transpose(tgt, (1, 2, 0))
    h, w, _ = tgt.shape
    img_h, img_w = (256, 352) if args.gt_type == "NYU" else (args.img_height, args.img_width)   # NYU size hard-coded (:154-155)
    if (not args.no_resize) and (h != img_h or w != img_w):
        tgt = KE.imresize_bilinear(tgt, (img_h, img_w)).astype(np.float32)
    t = torch.from_numpy(np.ascontiguousarray(np.transpose(tgt, (2, 0, 1)))).float()
    mean, std = normalization(args.imagenet_normalization, args.monodepth2)
    mean, std = torch.tensor(mean).view(3, 1, 1), torch.tensor(std).view(3, 1, 1)
    if args.gt_type == "KITTI":
        t = t / 255
    t = ((t - mean) / std).unsqueeze(0).to(device)
    if args.network in ("DORN", "disp_vgg_BN_DORN"):
        pred_d, _ = disp_net(t)
        pred_depth = torch.squeeze(U.get_depth_sid(pred_d, ordinal_c=args.ordinal_c, dataset=args.gt_type)).cpu().numpy()
    else:
        pred_depth = 1 / disp_net(t).cpu().numpy()[0, 0]
    gt = sample["gt_depth"]
    if args.gt_type == "NYU" and gt.ndim == 3:
        gt = gt[0]
    zoomed = zoom(pred_depth, (gt.shape[0] / pred_depth.shape[0], gt.shape[1] / pred_depth.shape[1])).clip(min_depth, max_depth)
    mask = sample["mask"] if args.gt_type == "KITTI" else (gt > min_depth) & (gt < max_depth)
    pz, g = zoomed[mask], gt[mask]
    if args.error and args.gt_type == "KITTI":
        worst_pixel_report(args, sample, gt, zoomed, g, pz, min_depth, max_depth, KE)
    if args.unsupervised or args.mono:
        scale = np.median(g) / np.median(pz)
    elif args.stereo:
        scale = 5.4
    else:
        scale = 1
    return KE.compute_errors(g, pz * scale), pred_depth


def worst_pixel_report(args, sample, gt, zoomed, g, pz, min_depth, max_depth, KE, out_root="output"):
    """test_disp.py:309-373 (--error): abs-rel per pixel of the zoomed, clipped prediction (x5.4 with --stereo, x median ratio with
    --mono; the reference leaves the map undefined -- NameError -- without either flag, here the unscaled prediction is used), the 300
    worst pixels inside the Garg crop, input + annotated input saved as PNGs.  Returns (abs_rel map, graph_index)."""
    if args.stereo:
        scale = 5.4
    elif args.mono:
        scale = np.median(g) / np.median(pz)
    else:
        scale = 1.0
    m = KE.compute_abs_rel_per_pixel(gt, zoomed * scale, min_depth=min_depth, max_depth=max_depth)
    graph_index, _ = KE.worst_pixels(m, 300)
    annotated = KE.annotate_pixels(sample["tgt"], graph_index)
    sub = "stereo" if args.stereo else ("mono" if args.mono else "")
    d = os.path.join(out_root, sub, "bad_300pixel")
    os.makedirs(d, exist_ok=True)
    j = sample.get("index", 0)
    from PIL import Image
    for name, arr in (("input", sample["tgt"]), ("annotate", annotated)):
        Image.fromarray(np.clip(arr, 0, 255).astype(np.uint8)).save(os.path.join(d, "{}_{}.png".format(j, name)))
    return m, graph_index


@torch.no_grad()
def main(argv=None):
    args = build_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("test_disp.py drives the MI355X HIP path; no GPU is visible (there is no CPU fallback)")
    device = torch.device("cuda")
    import __graft_entry__
    __graft_entry__.build(only_library=True)
    import supervised_dispnet_amd.models as models
    import supervised_dispnet_amd.networks as networks
    import supervised_dispnet_amd.utils as U
    from supervised_dispnet_amd import engine, kitti_eval as KE
    if args.compute is not None:
        engine.set_compute(args.compute)
    if args.gt_type not in ("KITTI", "NYU"):
        raise ValueError("gt-type '{}' is outside this path (KITTI and NYU are supported)".format(args.gt_type))
    if args.pretrained_posenet is not None:
        raise ValueError("PoseNet-scaled evaluation is outside this path; omit --pretrained-posenet")
    disp_net = create_disp_net(args, models, networks, device, U)
    if not (args.mono or args.stereo):
        disp_net.load_state_dict(torch.load(args.pretrained_dispnet, map_location=device)["state_dict"])
    disp_net.eval()
    print("no PoseNet specified, scale_factor will be determined by median ratio, which is kiiinda cheating "
          "(but consistent with original paper)")
    if args.gt_type == "KITTI":
        min_depth, max_depth = 1e-3, 80
        if args.dataset_list is not None:
            with open(args.dataset_list) as f:
                test_files = f.read().splitlines()
        else:
            test_files = sorted(n for n in os.listdir(args.dataset_dir) if n.split(".")[-1] in args.img_exts)
        framework = KE.KittiTestFramework(args.dataset_dir, test_files, min_depth=min_depth, max_depth=max_depth)
    else:
        min_depth, max_depth = 1e-3, 10
        framework = KE.NyuTestFramework(args.dataset_dir, min_depth=min_depth, max_depth=max_depth)
    n = len(framework)
    print("{} files to test".format(n))
    errors = np.zeros((7, n), np.float32)
    predictions = None
    for j in range(n):
        sample = framework[j]
        if isinstance(sample, dict):
            sample.setdefault("index", j)
        errs, pred_depth = evaluate_sample(args, disp_net, sample, device, min_depth, max_depth, KE, U)
        errors[:, j] = errs
        if args.output_dir is not None:
            if predictions is None:
                predictions = np.zeros((n,) + pred_depth.shape)
            predictions[j] = pred_depth
    mean_errors = errors.mean(1)
    names = ["abs_rel", "sq_rel", "rms", "log_rms", "a1", "a2", "a3"]
    print("Results with scale factor determined by GT/prediction ratio (like the original paper) : ")
    print("{:>10}, {:>10}, {:>10}, {:>10}, {:>10}, {:>10}, {:>10}".format(*names))
    print("&{:10.3f}& {:10.3f}& {:10.3f}& {:10.3f}& {:10.3f}& {:10.3f}& {:10.3f}".format(*mean_errors))
    if args.output_dir is not None:
        os.makedirs(args.output_dir, exist_ok=True)
        np.save(os.path.join(args.output_dir, "predictions.npy"), predictions)
    return mean_errors


if __name__ == "__main__":
    main()
