#!/usr/bin/env python3
"""Trainer for the DispNet hot path on MI355X -- the command line of the reference's train.py (flags, defaults and `dest`s of
train.py:30-91; loss / network selectors of :239-262,449-468; per-iteration sequence of :420-522; checkpoint keys of
:376-388), driving the HIP engine instead of cuDNN:

    python3 train.py DATA -b32 --network disp_vgg_BN --loss L1 --with-gt
    python3 -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py DATA -b32 ...   # one rank per GPU

What differs from the reference, on purpose:
  * one process per GPU over RCCL instead of single-process nn.DataParallel: `-b` stays the GLOBAL batch, every rank takes
    its contiguous slice, gradients are summed by the bucketed all-reduce (distributed.GradReducer), BatchNorm statistics
    stay per replica exactly as under DataParallel;
  * Adam runs as one fused kernel over a flat arena; the 123.6 M unused VGG-classifier parameters are not optimised;
  * --unsupervised gets the 5-tuple loader the reference's branch needs (C-1); unknown selectors raise ValueError (C-12);
  * `--synthetic N` (extension) trains on N synthetic samples with the statistics of SURVEY.md section 8d, for boxes
    without a dataset; tensorboard / progress-bar logging is replaced by plain prints and the two CSV logs.
"""
import argparse
import csv
import datetime
import os
import shutil
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

NETWORKS = ("dispnet", "disp_res", "disp_res_50", "disp_res_18", "disp_vgg", "disp_vgg_BN", "FCRN", "res50_aspp", "ASPP",
            "disp_res_101", "DORN", "disp_vgg_BN_DORN")
LOSSES = ("Multi_L1", "Multi_full_L1", "Multi_berhu", "Multi_L2", "L1", "berhu", "L2", "scale_inv", "Multi_scale_inv", "DORN")


def build_parser():
    p = argparse.ArgumentParser(description="Structure from Motion Learner training on KITTI and CityScapes Dataset",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--network", default="disp_vgg", type=str, help="network type")
    p.add_argument("--dataset", default="kitti", type=str, help="dataset name")
    p.add_argument("--imagenet-normalization", action="store_true", help="use imagenet parameter for normalization.")
    p.add_argument("--pretrained-encoder", action="store_true", help="use imagenet pretrained parameter.")
    p.add_argument("--loss", default="Multi_L1", type=str, help="loss type")
    p.add_argument("--ordinal-c", default=80, type=int, metavar="N", help="DORN loss channel number")
    p.add_argument("--diff-lr", action="store_true", help="use different learning rate for encoder and decoder")
    p.add_argument("--sgd", action="store_true", help="use sgd optimizer, if not then adam")
    p.add_argument("--record", action="store_true", help="save every epoch checkpoints")
    p.add_argument("--unsupervised", action="store_true", help="to have unsupervised loss")
    p.add_argument("--data-amount", default=1, type=float, metavar="M", help="percentage of data to be trained")
    p.add_argument("--monodepth2", action="store_true", help="to finetune over monodepth2 model")
    p.add_argument("data", metavar="DIR", help="path to dataset")
    p.add_argument("--dataset-format", default="sequential", metavar="STR", help="dataset format: stacked | sequential")
    p.add_argument("--sequence-length", type=int, metavar="N", help="sequence length for training", default=3)
    p.add_argument("--rotation-mode", type=str, choices=["euler", "quat"], default="euler", help="rotation mode for PoseExpnet")
    p.add_argument("--padding-mode", type=str, choices=["zeros", "border"], default="zeros", help="padding mode for image warping")
    p.add_argument("--with-gt", action="store_true", help="use ground truth for validation")
    p.add_argument("-j", "--workers", default=4, type=int, metavar="N", help="number of data loading workers")
    p.add_argument("--epochs", default=200, type=int, metavar="N", help="number of total epochs to run")
    p.add_argument("--epoch-size", default=0, type=int, metavar="N", help="manual epoch size (will match dataset size if not set)")
    p.add_argument("-b", "--batch-size", default=4, type=int, metavar="N", help="mini-batch size (global, over all ranks)")
    p.add_argument("--lr", "--learning-rate", default=1e-4, type=float, metavar="LR", help="initial learning rate")
    p.add_argument("--momentum", default=0.9, type=float, metavar="M", help="momentum for sgd, alpha parameter for adam")
    p.add_argument("--beta", default=0.999, type=float, metavar="M", help="beta parameters for adam")
    p.add_argument("--weight-decay", "--wd", default=0, type=float, metavar="W", help="weight decay")
    p.add_argument("--print-freq", default=10, type=int, metavar="N", help="print frequency")
    p.add_argument("-e", "--evaluate", dest="evaluate", action="store_true", help="evaluate model on validation set")
    p.add_argument("--pretrained-disp", dest="pretrained_disp", default=None, metavar="PATH", help="path to pre-trained dispnet model")
    p.add_argument("--pretrained-exppose", dest="pretrained_exp_pose", default=None, metavar="PATH",
                   help="path to pre-trained Exp Pose net model")
    p.add_argument("--seed", default=0, type=int, help="seed for random functions, and network initialization")
    p.add_argument("--log-summary", default="progress_log_summary.csv", metavar="PATH", help="csv of per-epoch train and valid stats")
    p.add_argument("--log-full", default="progress_log_full.csv", metavar="PATH", help="csv of per-gradient descent train stats")
    p.add_argument("-p", "--photo-loss-weight", type=float, help="weight for photometric loss", metavar="W", default=1)
    p.add_argument("-m", "--mask-loss-weight", type=float, help="weight for explainabilty mask loss", metavar="W", default=0)
    p.add_argument("-s", "--smooth-loss-weight", type=float, help="weight for disparity smoothness loss", metavar="W", default=0)
    p.add_argument("--log-output", action="store_true", help="will log dispnet outputs and warped imgs at validation step")
    p.add_argument("-f", "--training-output-freq", type=int, metavar="N", default=0,
                   help="frequence for outputting dispnet outputs and warped imgs at training")
    # extensions (not in the reference)
    p.add_argument("--synthetic", type=int, default=0, metavar="N", help="train on N synthetic samples instead of DIR")
    p.add_argument("--shards", default=None, metavar="SHARD_DIR",
                   help="training samples from pre-decoded uint8 shards (tools/make_shards.py DIR SHARD_DIR): flip / /255 / normalise run "
                        "on the GPU (dn_u8_normalize_flip), bit-identical to the JPEG loader's host chain")
    p.add_argument("--img-height", type=int, default=128, help="synthetic image height")
    p.add_argument("--img-width", type=int, default=416, help="synthetic image width")
    p.add_argument("--train-pose", action="store_true", help="also optimise PoseExpNet (the reference never does, appendix C-3)")
    p.add_argument("--tape", dest="tape", action="store_true", default=None,
                   help="supervised per-sample / multi-scale losses with the fused Adam: record the launches of one training step and re-issue "
                        "them with one call per step (supervised_dispnet_amd.graph.TapedStep); the second batch checks the replay against "
                        "the eager step bit for bit, a mismatch falls back to eager launches with a message.  This is the DEFAULT wherever the "
                        "setting is eligible (at the reference's own -b4 the eager step is host-bound: 5.0 ms against 3.5 taped); an "
                        "ineligible setting runs eagerly without a word unless --tape was given explicitly")
    p.add_argument("--no-tape", dest="tape", action="store_false", help="always issue the training step as eager launches")
    p.add_argument("--save-root", default="checkpoints", help="directory under which the run folder is created")
    p.add_argument("--legacy-align-corners", action="store_true",
                   help="photometric warp with F.grid_sample(align_corners=True): the sampling the reference's pinned torch 1.0.1 did at "
                        "inverse_warp.py:191 (current torch, and this path by default, sample with align_corners=False; SURVEY.md section 5)")
    p.add_argument("--compute", choices=["f32x3", "f32", "bf16"], default=None,
                   help="arithmetic of the matrix-core convolution kernels (tensors, statistics, losses and Adam are fp32 in every mode): "
                        "f32x3 (library default) fp32 products from three exact bf16 pieces per operand; f32 the fp32 matrix instruction; "
                        "bf16 operands ROUNDED to bf16 with fp32 accumulation = the mixed-precision run of BASELINE configs[4]")
    return p


def save_path_formatter(args, parser):
    """utils.py:11-43: DIR name, then every non-default key of a fixed list, then a timestamp."""
    d = vars(args)
    parts = [os.path.basename(os.path.normpath(d["data"]))]
    if d["epochs"] != parser.get_default("epochs"):
        parts.append("{}epochs".format(d["epochs"]))
    for key, prefix in (("epoch_size", "epoch_size"), ("sequence_length", "seq"), ("rotation_mode", "rot_"), ("padding_mode", "padding_"),
                        ("batch_size", "b"), ("lr", "lr"), ("photo_loss_weight", "p"), ("mask_loss_weight", "m"),
                        ("smooth_loss_weight", "s"), ("network", "network"), ("pretrained_encoder", "pretrained_encoder"), ("loss", "loss")):
        if d[key] != parser.get_default(key):
            parts.append("{}{}".format(prefix, d[key]))
    return os.path.join(",".join(parts), datetime.datetime.now().strftime("%m-%d-%H:%M"))


def save_checkpoint(save_path, dispnet_state, exp_pose_state, is_best, epoch, filename="checkpoint.pth.tar", record=False):
    """utils.py:85-99 file names and dict keys."""
    for prefix, state in (("dispnet", dispnet_state), ("exp_pose", exp_pose_state)):
        torch.save(state, os.path.join(save_path, "{}_{}".format(prefix, filename)))
    if record:
        rec = os.path.join(save_path, "weights_{}".format(epoch))
        os.makedirs(rec, exist_ok=True)
        torch.save(dispnet_state, os.path.join(rec, "dispnet_{}".format(filename)))
    if is_best:
        for prefix in ("dispnet", "exp_pose"):
            shutil.copyfile(os.path.join(save_path, "{}_{}".format(prefix, filename)),
                            os.path.join(save_path, "{}_model_best.pth.tar".format(prefix)))


def create_disp_net(args, models, networks, device, U=None):
    if args.monodepth2:
        if args.network == "disp_vgg_BN":
            enc = networks.vggEncoder(num_layers=16, pretrained=False).to(device)
        elif args.network == "disp_res_18":
            enc = networks.ResnetEncoder(num_layers=18, pretrained=False).to(device)
        else:
            raise ValueError("undefined network")
        dec = networks.DepthDecoder(enc.num_ch_enc).to(device)
        # "when monodepth2, it must load existing weight (not include adam)" -- train.py:236-237
        U.load_model({"encoder": enc, "depth": dec}, args.pretrained_disp)
        return models.monodepth2(encoder=enc, decoder=dec)
    table = {"dispnet": "DispNetS", "disp_res": "Disp_res", "disp_res_50": "Disp_res_50", "disp_res_18": "Disp_res_18",
             "disp_vgg": "Disp_vgg_feature", "disp_vgg_BN": "Disp_vgg_BN", "FCRN": "FCRN", "res50_aspp": "res50_aspp",
             "ASPP": "deeplab_depth", "disp_res_101": "Disp_res_101"}
    if args.network in table:
        return getattr(models, table[args.network])(datasets=args.dataset).to(device)
    if args.network == "DORN":
        return models.DORN(freeze=args.diff_lr, datasets=args.dataset).to(device)
    if args.network == "disp_vgg_BN_DORN":
        return models.Disp_vgg_BN_DORN(ordinal_c=args.ordinal_c, datasets=args.dataset).to(device)
    raise ValueError("undefined network")


def supervised_loss(args, LF, gt_depth, depth, pred_ord=None, target_c=None):
    L = args.loss
    if L == "Multi_L1":
        return LF.Multiscale_L1_loss(gt_depth, depth)
    if L == "Multi_full_L1":
        return LF.Multiscale_FULL_L1_loss(gt_depth, depth)
    if L == "Multi_berhu":
        return LF.Multiscale_berhu_loss(gt_depth, depth)
    if L == "Multi_L2":
        return LF.Multiscale_L2_loss(gt_depth, depth)
    if L == "L1":
        return LF.l1_loss(gt_depth, depth, args.dataset)
    if L == "berhu":
        return LF.berhu_loss(gt_depth, depth, args.dataset)
    if L == "L2":
        return LF.l2_loss(gt_depth, depth, args.dataset)
    if L == "scale_inv":
        return LF.Scale_invariant_loss(gt_depth, depth, args.dataset)
    if L == "Multi_scale_inv":
        return LF.Multiscale_scale_inv_loss(gt_depth, depth)
    if L == "DORN":
        return LF.DORN_loss(gt_depth, pred_ord, target_c, args.dataset)
    raise ValueError("undefined loss")


class Meter(object):
    def __init__(self, n=1):
        self.sum, self.count, self.val = np.zeros(n), 0, np.zeros(n)

    def update(self, v, k=1):
        v = np.atleast_1d(np.asarray(v, dtype=np.float64))
        self.val = v
        self.sum += v * k
        self.count += k

    @property
    def avg(self):
        return self.sum / max(self.count, 1)


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("train.py drives the MI355X HIP path; no GPU is visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)
    if args.batch_size % world != 0:
        raise SystemExit("-b %d must divide over %d ranks" % (args.batch_size, world))

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build(only_library=True)
    if world > 1:
        dist.barrier()
    import supervised_dispnet_amd.loss_functions as LF
    import supervised_dispnet_amd.models as models
    import supervised_dispnet_amd.networks as networks
    import supervised_dispnet_amd.utils as U
    from supervised_dispnet_amd import data as D, engine
    from supervised_dispnet_amd.distributed import GradReducer
    from supervised_dispnet_amd.functional import reciprocal
    from supervised_dispnet_amd.optim import FusedAdam

    if args.compute is not None:
        engine.set_compute(args.compute)
    save_path = os.path.join(args.save_root, save_path_formatter(args, parser))
    if rank == 0:
        print("=> will save everything to {}".format(save_path))
        print("=> matrix-core arithmetic: {}{}".format(engine.compute_mode(), ", warp sampling align_corners=True" if args.legacy_align_corners else ""))
        os.makedirs(save_path, exist_ok=True)
    torch.manual_seed(args.seed)
    if args.evaluate:
        args.epochs = 0

    # ---- data
    with_refs = bool(args.unsupervised)
    if args.synthetic > 0:
        maxd = 10.0 if args.dataset == "nyu" else 80.0
        train_set = D.SyntheticDepthSet(args.synthetic, args.img_height, args.img_width, max_depth=maxd, seed=args.seed,
                                        sequence_length=args.sequence_length, with_refs=with_refs)
        val_set = D.SyntheticDepthSet(max(args.batch_size, args.synthetic // 4), args.img_height, args.img_width, max_depth=maxd,
                                      seed=args.seed + 1, sequence_length=args.sequence_length, with_refs=with_refs and not args.with_gt)
    elif args.dataset == "kitti":
        if args.dataset_format != "sequential":
            raise ValueError("only the sequential folder format is supported on this path")
        mean, std = D.normalization(args.imagenet_normalization, args.monodepth2)
        print("=> fetching scenes in '{}'".format(args.data))
        if args.shards:
            train_set = None                         # --shards replaces the JPEG train set: do not crawl the scene folders for it
        else:
            train_set = D.SequenceFolder(args.data, transform=D.Transform(mean, std, flip=True), seed=args.seed, train=True,
                                         sequence_length=args.sequence_length, percentage=args.data_amount, with_refs=with_refs)
        if args.with_gt:
            val_set = D.ValidationSet(args.data, transform=D.Transform(mean, std, flip=False))
        else:
            val_set = D.SequenceFolder(args.data, transform=D.Transform(mean, std, flip=False), seed=args.seed, train=False,
                                       sequence_length=args.sequence_length, with_refs=True)
    else:
        raise ValueError("dataset '{}' needs the reference's NYU h5 loader, which is outside this path; use --synthetic".format(args.dataset))
    per_rank = args.batch_size // world
    workers = min(per_rank, os.cpu_count() or 1)     # the reference uses num_workers = batch_size and ignores -j (train.py:201-206)
    if args.shards:
        from supervised_dispnet_amd.shards import ShardLoader
        mean, std = D.normalization(args.imagenet_normalization, args.monodepth2)
        train_loader = ShardLoader(args.shards, per_rank, device, mean=mean, std=std, flip=True, shuffle=True, seed=args.seed, rank=rank,
                                   world=world, with_refs=with_refs, percentage=args.data_amount)
        train_sampler = train_loader                 # set_epoch() reshuffles (and re-seeds the flip draws)
        if rank == 0:
            print("{} samples found in {} train scenes (shards)".format(len(train_loader.set), len(train_loader.set.scenes)))
    else:
        if rank == 0:
            print("{} samples found in {} train scenes".format(len(train_set), len(train_set.scenes)))
        train_sampler = D.RankSampler(len(train_set), args.batch_size, rank, world, shuffle=True, seed=args.seed)
        train_loader = torch.utils.data.DataLoader(train_set, batch_sampler=train_sampler, num_workers=workers, pin_memory=True)
    if rank == 0:
        print("{} samples found in {} valid scenes".format(len(val_set), len(val_set.scenes)))
    val_sampler = D.RankSampler(len(val_set), args.batch_size, rank, world, shuffle=False, drop_last=False)
    val_loader = torch.utils.data.DataLoader(val_set, batch_sampler=val_sampler, num_workers=workers, pin_memory=True)
    if args.epoch_size == 0:
        args.epoch_size = len(train_loader)

    # ---- models
    if rank == 0:
        print("=> creating model")
    disp_net = create_disp_net(args, models, networks, device, U)
    output_exp = args.mask_loss_weight > 0
    pose_exp_net = models.PoseExpNet(nb_ref_imgs=args.sequence_length - 1, output_exp=output_exp).to(device)
    if args.pretrained_exp_pose:
        pose_exp_net.load_state_dict(torch.load(args.pretrained_exp_pose, map_location=device)["state_dict"], strict=False)
    else:
        pose_exp_net.init_weights()
    weights = None
    if args.pretrained_disp and not args.monodepth2:
        weights = torch.load(args.pretrained_disp, map_location=device)
        disp_net.load_state_dict(weights["state_dict"])
    elif not args.monodepth2:
        disp_net.init_weights(use_pretrained_weights=args.pretrained_encoder)

    hot = list(disp_net._hot_parameters()) if hasattr(disp_net, "_hot_parameters") else [p for p in disp_net.parameters() if p.requires_grad]
    if args.train_pose:
        hot += [p for p in pose_exp_net.parameters() if p.requires_grad]
    reducer = None
    plain_params = None          # torch.optim path (--sgd / --diff-lr): gradients live in p.grad and are exchanged per step
    if args.diff_lr:
        if not hasattr(disp_net, "get_1x_lr_params"):
            # the reference's only network with these methods is models/DORN.py:222-236 (train.py:306-312 calls them unconditionally and
            # dies with AttributeError for every other --network); the full DORN model is outside SURVEY.md section 8
            raise SystemExit("--diff-lr needs a network with get_1x_lr_params() / get_10x_lr_params() (reference: models.DORN only)")
        groups = [{"params": disp_net.get_1x_lr_params(), "lr": args.lr}, {"params": disp_net.get_10x_lr_params(), "lr": args.lr * 10}]
        optimizer = torch.optim.SGD(groups, lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay)
    elif args.sgd:
        optimizer = torch.optim.SGD([{"params": hot, "lr": args.lr}], lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay)
    if args.diff_lr or args.sgd:
        plain_params = [p for g in optimizer.param_groups for p in g["params"]]
    else:
        if rank == 0:
            print("=> setting adam solver")
        order = disp_net._grad_production_order() if hasattr(disp_net, "_grad_production_order") else None
        optimizer = FusedAdam(hot, lr=args.lr, betas=(args.momentum, args.beta), weight_decay=args.weight_decay, production_order=order)
        if world > 1:
            reducer = GradReducer(optimizer.arena)
            engine.GradSink.reducer = reducer
    if weights is not None and "optimizer" in weights and not args.monodepth2:
        try:
            optimizer.load_state_dict(weights["optimizer"])
        except Exception as e:    # a reference checkpoint carries torch.optim.Adam state (incl. the unused classifier slots)
            print("=> optimizer state not restored ({}); Adam moments start from zero".format(e))

    if rank == 0:
        with open(os.path.join(save_path, args.log_summary), "w") as f:
            csv.writer(f, delimiter="\t").writerow(["train_loss", "validation_loss"])
        with open(os.path.join(save_path, args.log_full), "w") as f:
            csv.writer(f, delimiter="\t").writerow(["train_loss", "photo_loss", "explainability_loss", "smooth_loss"])

    ctx = dict(args=args, device=device, LF=LF, U=U, reciprocal=reciprocal, rank=rank, world=world, reducer=reducer, save_path=save_path,
               plain_params=plain_params)
    if args.tape is not False and device.type == "cuda":
        ctx["tape"] = TapedTrainer(args, LF, reciprocal, disp_net, optimizer, reducer, rank, explicit=args.tape is True)

    def run_validation(epoch):
        if args.with_gt:
            return validate_with_gt(ctx, val_loader, disp_net)
        return validate_without_gt(ctx, val_loader, disp_net, pose_exp_net)

    if args.pretrained_disp or args.evaluate:
        errors, names = run_validation(0)
        if rank == 0:
            print(" * Avg " + ", ".join("{} : {:.3f}".format(n, e) for n, e in zip(names, errors)))

    best_error, n_iter = -1.0, 0
    for epoch in range(args.epochs):
        train_sampler.set_epoch(epoch)
        train_loss, n_iter = train(ctx, train_loader, disp_net, pose_exp_net, optimizer, args.epoch_size, n_iter)
        if rank == 0:
            print(" * Avg Loss : {:.3f}".format(train_loss))
        errors, names = run_validation(epoch)
        if rank == 0:
            print(" * Avg " + ", ".join("{} : {:.3f}".format(n, e) for n, e in zip(names, errors)))
            decisive = float(errors[1])
            if best_error < 0:
                best_error = decisive
            is_best = decisive < best_error
            best_error = min(best_error, decisive)
            save_checkpoint(save_path,
                            {"epoch": epoch + 1, "state_dict": disp_net.state_dict(), "optimizer": optimizer.state_dict()},
                            {"epoch": epoch + 1, "state_dict": pose_exp_net.state_dict()}, is_best, epoch, record=args.record)
            with open(os.path.join(save_path, args.log_summary), "a") as f:
                csv.writer(f, delimiter="\t").writerow([train_loss, decisive])
    if ctx.get("tape") is not None:
        ctx["tape"].close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _to_device_batch(batch, device, unsupervised):
    if unsupervised:
        tgt, refs, intr, intr_inv, gt = batch
        return (tgt.to(device, non_blocking=True), [r.to(device, non_blocking=True) for r in refs],
                torch.as_tensor(intr).float().to(device), torch.as_tensor(intr_inv).float().to(device), gt.to(device, non_blocking=True))
    tgt, gt = batch
    return tgt.to(device, non_blocking=True), None, None, None, gt.to(device, non_blocking=True)


class TapedTrainer(object):
    """--tape: the supervised training step (forward, 1/disp, loss, backward, gradient exchange, Adam) through a launch tape.
    The first full-size batch is recorded (a real step); the second one is written into the tape's static input buffers and the replay
    is checked against the same step issued eagerly from the same state, bit for bit -- on a batch the tape has not seen, so work it
    missed would show up as stale data; only then are further batches replayed.  Batches of another shape (the last one of an epoch),
    a failed recording or a failed check run eagerly."""

    def __init__(self, args, LF, reciprocal, disp_net, optimizer, reducer, rank, explicit=True):
        self.explicit = explicit      # --tape given: say why an ineligible setting runs eagerly (the default decides silently)
        self.args, self.LF, self.reciprocal = args, LF, reciprocal
        self.net, self.opt, self.reducer, self.rank = disp_net, optimizer, reducer, rank
        self.ts = None
        self.state = "new"            # new -> recorded -> verified | off
        self.img = self.gt = None
        why = None
        if args.unsupervised or args.loss == "DORN" or args.monodepth2:
            why = "only the supervised depth losses are taped"
        elif args.photo_loss_weight != 1 or args.mask_loss_weight != 0 or args.smooth_loss_weight != 0:
            why = "needs -p 1 -m 0 -s 0 (the weighted sum of loss terms is framework-side arithmetic)"
        elif not hasattr(optimizer, "arena"):
            why = "needs the fused Adam (not --sgd / --diff-lr)"
        else:
            from supervised_dispnet_amd.distributed import data_parallel_world
            if data_parallel_world() > 1 and (args.loss.startswith("Multi") or args.loss == "DORN"):
                # decided up front: found out by a failed recording, the first batch's forward ran twice (BatchNorm
                # running statistics updated twice) before the eager fallback took over
                why = "--loss {} exchanges whole-batch statistics between the ranks inside the forward pass (a host-side collective)".format(args.loss)
        # every rank takes the same decision (one rank on the tape and another eager would interleave their collectives differently)
        if not self._agree(why is None) and why is None:
            why = "another rank cannot use the tape"
        if why is not None:
            self._off(why, quiet=not explicit)

    def _agree(self, ok):
        from supervised_dispnet_amd.distributed import agree_all_ranks
        dev = next(self.net.parameters()).device
        return agree_all_ranks(ok, device=dev if dev.type == "cuda" else None)

    def close(self):
        """End of training: free the tape now (graph.TapedStep.close -- a DataLoader worker forked later must not inherit it as garbage)."""
        if self.ts is not None:
            self.ts.close()
            self.ts = None
        self.state = "off"

    def _off(self, why, quiet=False):
        self.state = "off"
        if self.rank == 0 and not quiet:
            print("=> --tape: eager launches ({})".format(why))

    def usable(self, img, gt):
        if self.state == "off":
            return False
        return self.img is None or (img.shape == self.img.shape and gt.shape == self.gt.shape)

    def _step(self):
        from supervised_dispnet_amd.graph import backward
        depth = [self.reciprocal(d) for d in self.net(self.img)]
        loss = supervised_loss(self.args, self.LF, self.gt, depth)
        self.opt.zero_grad()
        backward(loss)
        self.opt.step(grad_scale=self.reducer.finish() if self.reducer is not None else 1.0)
        return loss

    def step(self, img, gt):
        """One training step on (img, gt); returns the loss value, or None when the caller has to run the step eagerly."""
        from supervised_dispnet_amd.graph import TapedStep
        if self.img is None:
            self.img, self.gt = img.clone(), gt.clone()
        else:
            self.img.copy_(img)
            self.gt.copy_(gt)
        agreed = False                 # this call's agreement collective has completed (never issue a second, unmatched one)
        try:
            if self.state == "new":
                self.ts = TapedStep(self._step, optimizer=self.opt, warmup=0, static_inputs=(self.img, self.gt))
                loss = self.ts()                                   # recorded = executed
                self.state = "recorded"
                ok = self._agree(True)
                agreed = True
                if not ok:
                    self._off("the recording failed on another rank")
                return float(loss.item())
            if self.state == "recorded":
                st = [self.opt.arena.flat_p, self.opt.exp_avg, self.opt.exp_avg_sq, self.opt._dev["step"], self.opt._dev["derived"]]
                st += [b for b in self.net.buffers() if b.is_cuda]
                same, worst = self.ts.verify(st)                   # (leaves the state one step further: this batch's eager step)
                ok = self._agree(bool(same))
                agreed = True
                if same and not ok:
                    same, worst = False, float("nan")              # (another rank's check failed: all ranks leave the tape together)
                if same:
                    self.state = "verified"
                    if self.rank == 0:
                        print("=> --tape: {} launches + {} stream fences per step, {} segment(s); replay == eager step on the second "
                              "batch, bit for bit".format(self.ts.launches, self.ts.fences, self.ts.segments))
                else:
                    self._off("replay differs from the eager step on the second batch, max |diff| {:.3g}".format(worst))
                return float(self.ts.last_eager_out[0].item())     # (either way this batch's step was taken: the check's eager one)
            return float(self.ts().item())
        except Exception as e:          # noqa: BLE001 -- a setting the tape refuses
            if self.state == "verified":
                raise
            from supervised_dispnet_amd.distributed import data_parallel_world
            if data_parallel_world() > 1:
                # Beyond one rank a failed recording cannot be retried eagerly by THIS rank alone: its peers have taken the step (their
                # all-reduces are issued), so an eager re-run here would pair this step's buckets with their next step's.  Tell the peers
                # (if they still wait in this call's agreement) and stop the job.
                if not agreed:
                    self._agree(False)
                raise RuntimeError("--tape: recording / check failed on rank {} of a data-parallel run ({}: {}); restart without --tape"
                                   .format(self.rank, type(e).__name__, str(e)[:160])) from e
            self._off("{}: {}".format(type(e).__name__, str(e)[:160]))
            return None


def train(ctx, loader, disp_net, pose_exp_net, optimizer, epoch_size, n_iter):
    args, device, LF, U, reciprocal = ctx["args"], ctx["device"], ctx["LF"], ctx["U"], ctx["reciprocal"]
    rank, reducer = ctx["rank"], ctx["reducer"]
    w1, w2, w3 = args.photo_loss_weight, args.mask_loss_weight, args.smooth_loss_weight
    disp_net.train()
    pose_exp_net.train()
    if args.diff_lr:          # freeze BN (train.py:406-411)
        for m in disp_net.modules():
            if m.__class__.__name__.find("BatchNorm") != -1:
                m.eval()
    losses, batch_time, data_time = Meter(), Meter(), Meter()
    end = time.time()
    tape = ctx.get("tape")
    for i, batch in enumerate(loader):
        data_time.update(time.time() - end)
        tgt_img, ref_imgs, intrinsics, intrinsics_inv, gt_depth = _to_device_batch(batch, device, args.unsupervised)
        if tape is not None and tape.usable(tgt_img, gt_depth):
            lv = tape.step(tgt_img, gt_depth)
            if lv is not None:
                losses.update(lv, args.batch_size)
                batch_time.update(time.time() - end)
                end = time.time()
                if rank == 0:
                    with open(os.path.join(ctx["save_path"], args.log_full), "a") as f:
                        csv.writer(f, delimiter="\t").writerow([lv, lv, 0, 0])
                    if i % args.print_freq == 0:
                        print("Train: [{}/{}] Time {:.3f} ({:.3f}) Data {:.3f} Loss {:.4f} ({:.4f})".format(
                            i, min(len(loader), epoch_size), batch_time.val[0], batch_time.avg[0], data_time.avg[0], lv, losses.avg[0]))
                if i >= epoch_size - 1:
                    break
                n_iter += 1
                continue
        explainability_mask, pose = (None, None)
        if args.unsupervised:
            explainability_mask, pose = pose_exp_net(tgt_img, ref_imgs)
        if args.dataset == "nyu" and gt_depth.dim() == 4:
            gt_depth = torch.squeeze(gt_depth[:, 0, :, :])
        pred_ord = target_c = depth = None
        if args.loss == "DORN":
            target_c = U.get_labels_sid(gt_depth, ordinal_c=args.ordinal_c, dataset=args.dataset)
            _pred_d, pred_ord = disp_net(tgt_img)
        else:
            disparities = disp_net(tgt_img)
            if args.monodepth2:
                depth = [5.4 * reciprocal(d) for d in disparities]       # 5.4 = stereo scale factor (train.py:443)
            else:
                depth = [reciprocal(d) for d in disparities]
        if not args.unsupervised:
            loss_1 = supervised_loss(args, LF, gt_depth, depth, pred_ord, target_c)
        else:
            loss_1 = LF.photometric_reconstruction_loss(tgt_img, ref_imgs, intrinsics, intrinsics_inv, depth, explainability_mask, pose,
                                                        args.rotation_mode, args.padding_mode, align_corners=args.legacy_align_corners)
        loss_2 = LF.explainability_loss(explainability_mask) if w2 > 0 else 0
        # the reference evaluates the smoothness term unconditionally and multiplies it by -s (train.py:483-488);
        # with -s 0 (the README recipe) the product is exactly 0 for the finite values it takes, so it is skipped
        if w3 != 0:
            loss_3 = LF.smooth_DORN_loss(pred_ord) if args.loss == "DORN" else LF.smooth_loss(depth)
        else:
            loss_3 = 0
        loss = w1 * loss_1 + w2 * loss_2 + w3 * loss_3
        optimizer.zero_grad()
        loss.backward()
        if reducer is not None:
            optimizer.step(grad_scale=reducer.finish())
        else:
            if ctx["world"] > 1 and ctx["plain_params"] is not None:
                from supervised_dispnet_amd.distributed import average_plain_grads
                average_plain_grads(ctx["plain_params"])       # what DataParallel's reduce does for every optimizer (train.py:316)
            optimizer.step()
        lv = float(loss.item())
        losses.update(lv, args.batch_size)
        batch_time.update(time.time() - end)
        end = time.time()
        if rank == 0:
            with open(os.path.join(ctx["save_path"], args.log_full), "a") as f:
                csv.writer(f, delimiter="\t").writerow([lv, float(loss_1.item()), float(loss_2.item()) if w2 > 0 else 0,
                                                         float(loss_3.item()) if w3 != 0 else 0])
            if i % args.print_freq == 0:
                print("Train: [{}/{}] Time {:.3f} ({:.3f}) Data {:.3f} Loss {:.4f} ({:.4f})".format(
                    i, min(len(loader), epoch_size), batch_time.val[0], batch_time.avg[0], data_time.avg[0], lv, losses.avg[0]))
        if i >= epoch_size - 1:
            break
        n_iter += 1
    return float(losses.avg[0]), n_iter


@torch.no_grad()
def validate_with_gt(ctx, loader, disp_net):
    args, device, LF, U = ctx["args"], ctx["device"], ctx["LF"], ctx["U"]
    names = ["abs_diff", "abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3"]
    errors = Meter(len(names))
    disp_net.eval()
    for i, (tgt_img, depth) in enumerate(loader):
        tgt_img, depth = tgt_img.to(device), depth.to(device)
        if args.dataset == "nyu" and depth.dim() == 4:
            depth = torch.squeeze(depth[:, 0, :, :])
        if args.loss == "DORN":
            pred, _ = disp_net(tgt_img)
            output_depth = torch.squeeze(U.get_depth_sid(pred, ordinal_c=args.ordinal_c, dataset=args.dataset), 1)
        else:
            output_depth = 1 / disp_net(tgt_img)[:, 0]
            if args.monodepth2:
                output_depth = output_depth * 5.4
        if output_depth.shape[-2:] != depth.shape[-2:]:
            output_depth = torch.nn.functional.interpolate(output_depth.unsqueeze(1), size=depth.shape[-2:], mode="bilinear",
                                                           align_corners=True).squeeze(1)
        errors.update(LF.compute_errors(depth, output_depth, dataset=args.dataset, unsupervised=args.unsupervised), tgt_img.size(0))
    return _all_rank_average(ctx, errors), names


@torch.no_grad()
def validate_without_gt(ctx, loader, disp_net, pose_exp_net):
    args, device, LF = ctx["args"], ctx["device"], ctx["LF"]
    w1, w2, w3 = args.photo_loss_weight, args.mask_loss_weight, args.smooth_loss_weight
    losses = Meter(3)
    disp_net.eval()
    pose_exp_net.eval()
    for i, batch in enumerate(loader):
        tgt_img, ref_imgs, intrinsics, intrinsics_inv, _gt = _to_device_batch(batch, device, True)
        disp = disp_net(tgt_img)
        depth = 1 / disp
        explainability_mask, pose = pose_exp_net(tgt_img, ref_imgs)
        l1 = float(LF.photometric_reconstruction_loss(tgt_img, ref_imgs, intrinsics, intrinsics_inv, depth, explainability_mask, pose,
                                                      args.rotation_mode, args.padding_mode, align_corners=args.legacy_align_corners).item())
        l2 = float(LF.explainability_loss(explainability_mask).item()) if w2 > 0 else 0.0
        l3 = float(LF.smooth_loss(depth).item())
        losses.update([w1 * l1 + w2 * l2 + w3 * l3, l1, l2], tgt_img.size(0))
    return _all_rank_average(ctx, losses), ["Total loss", "Photo loss", "Exp loss"]


def _all_rank_average(ctx, meter):
    if ctx["world"] > 1:
        import torch.distributed as dist
        t = torch.tensor(np.r_[meter.sum, meter.count], dtype=torch.float64, device=ctx["device"])
        dist.all_reduce(t)
        return (t[:-1] / t[-1].clamp(min=1)).cpu().numpy()
    return meter.avg


if __name__ == "__main__":
    main()
