#!/usr/bin/env python3
"""Headline benchmark: images/sec, forward + backward (+ Adam step), Disp_vgg_BN, 128x416, batch 32 per GPU, synthetic
KITTI-like data resident in HBM (BASELINE.json metric / configs[1]).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One rank per GPU (RCCL over xGMI through torch.distributed "nccl"); weak scaling: every rank steps its own b32 batch and
the only exchange is the bucketed gradient all-reduce overlapped with the encoder backward.  Rank 0 prints ONE JSON line.

Besides the contract fields the line carries
  roofline     -- for the dominant kernel (by time): algorithmic FLOP/s = sum over its launches of 2*MACs of the DIRECT
                  convolution divided by the sum of its launch durations, measured with HIP events on
                  the launch stream during extra instrumented steps that follow the timed region (so the events do not
                  perturb `value`; they run single-stream, the timed steps put the weight gradients on a side stream);
                  peak = 157.3 TFLOP/s dense fp32 MFMA.  The Winograd F(2x2,3x3) kernels execute 2.25x
                  fewer multiply-accumulates than they are credited with, so their `frac` may exceed 1; `executed_frac` =
                  frac / 2.25 is the share of the matrix peak their MFMAs actually occupy.
  cpu_baseline -- the CPU oracle (oracle/, a PyTorch-CPU restatement pinned to the reference's golden vectors; kind "port")
                  running the same training step on this box's host cores, on a bounded sample (small batch, few steps).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

PEAK_FP32_MFMA_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix peak
TRAIN_GFLOP_PER_IMG = 110.54         # BASELINE.md section 2 (conv/convT MACs*2, fwd + dgrad + wgrad, no dgrad for layer 0)


def synthetic_batch(batch, h, w, device, seed):
    """U(0,1) image normalised (x-0.5)/0.5; 5 %-dense U(1,80) ground truth."""
    g = torch.Generator().manual_seed(seed)
    img = (torch.rand(batch, 3, h, w, generator=g) - 0.5) / 0.5
    depth = torch.rand(batch, h, w, generator=g) * 79.0 + 1.0
    mask = (torch.rand(batch, h, w, generator=g) < 0.05).float()
    return img.to(device), (depth * mask).to(device)


def cpu_baseline(h, w, batch, steps, warmup):
    """The oracle's Disp_vgg_BN + l1_loss + Adam training step on the host cores (bounded sample)."""
    from oracle import losses as OL, nets as ON
    # oneDNN/OpenMP on a 256-thread host oversubscribes badly at this problem size (measured 0.04 img/s with 256
    # threads); 32 threads is what the baseline actually uses and reports.
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    sd = ON.xavier_init_(ON.disp_vgg_bn_state_dict(), torch.Generator().manual_seed(0))
    params = []
    for k, v in sd.items():
        if torch.is_floating_point(v) and "running" not in k:
            v.requires_grad_(True)
            params.append(v)
    opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.999))
    img, gt = synthetic_batch(batch, h, w, "cpu", 0)
    t0 = None
    for it in range(warmup + steps):
        if it == warmup:
            t0 = time.perf_counter()
        depth = [1 / d for d in ON.disp_vgg_bn(sd, img, training=True)]
        loss = OL.l1_loss(gt, depth, "kitti")
        opt.zero_grad()
        loss.backward()
        opt.step()
    dt = time.perf_counter() - t0
    return {"value": batch * steps / dt, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "oracle Disp_vgg_BN+L1+Adam, %dx%d, batch %d, %d timed steps (%d warm-up), torch %d threads" % (
                h, w, batch, steps, warmup, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--width", type=int, default=416)
    ap.add_argument("--profile-steps", type=int, default=2, help="instrumented steps (after the timed region) for the roofline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-layer", action="store_true", help="print the per-layer launch table to stderr")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--cpu-steps", type=int, default=2)
    args = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # "nccl" IS RCCL on ROCm (xGMI); DN_DIST_BACKEND=gloo only exists to exercise this path with 2 ranks on ONE GPU
        backend = os.environ.get("DN_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build(only_library=True)
    if world > 1:
        dist.barrier()
    import supervised_dispnet_amd.loss_functions as LF
    import supervised_dispnet_amd.models as models
    from supervised_dispnet_amd import engine
    from supervised_dispnet_amd.distributed import GradReducer
    from supervised_dispnet_amd.functional import reciprocal
    from supervised_dispnet_amd.optim import FusedAdam

    torch.manual_seed(0)                                   # identical random-init replicas on every rank
    net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
    _quiet_init(net)
    net.to(dev).train()
    opt = FusedAdam(net._hot_parameters(), lr=1e-4, betas=(0.9, 0.999), production_order=net._grad_production_order())
    reducer = GradReducer(opt.arena) if world > 1 else None
    engine.GradSink.reducer = reducer
    img, gt = synthetic_batch(args.batch, args.height, args.width, dev, seed=rank)

    def step():
        disparities = net(img)
        depth = [reciprocal(d) for d in disparities]
        loss = LF.l1_loss(gt, depth, "kitti")               # README recipe: --loss L1 -s 0 (smoothness weight 0)
        opt.zero_grad()
        loss.backward()
        scale = reducer.finish() if reducer is not None else 1.0
        opt.step(grad_scale=scale)
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())

    # ---- instrumented steps (not part of `value`): HIP events around every implicit-GEMM launch
    roofline = None
    if args.profile_steps > 0 and rank != 0:
        for _ in range(args.profile_steps):      # the steps carry collectives: every rank takes them, rank 0 records
            step()
        torch.cuda.synchronize()
    if rank == 0 and args.profile_steps > 0:
        engine.PROFILE = []
        for _ in range(args.profile_steps):
            step()
        torch.cuda.synchronize()
        agg = {}
        if args.per_layer:
            seen = {}
            for name, flops, e0, e1, tag in engine.PROFILE:
                r = seen.setdefault((name, tag), [0.0, 0.0, 0])
                r[0] += flops; r[1] += e0.elapsed_time(e1) * 1e-3; r[2] += 1
            print("%-28s %-58s %9s %9s %8s" % ("kernel", "layer", "ms/launch", "GFLOP", "TFLOP/s"), file=sys.stderr)
            for (name, tag), (fl, sec, n) in seen.items():
                print("%-28s %-58s %9.3f %9.2f %8.1f" % (name, tag, sec / n * 1e3, fl / n / 1e9, fl / sec / 1e12), file=sys.stderr)
        for name, flops, e0, e1, _tag in engine.PROFILE:
            a = agg.setdefault(name, [0.0, 0.0, 0])
            a[0] += flops
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
        engine.PROFILE = None
        dom = max(agg.items(), key=lambda kv: kv[1][1])
        name, (fl, sec, n) = dom
        wino = "wino_" in name
        roofline = {"bound": "mfma", "kernel": name, "achieved": fl / sec / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": fl / sec / 1e12 / PEAK_FP32_MFMA_TFLOPS, "traffic": pmc_traffic(name),
                    "algorithm": "winograd F(2x2,3x3): 16/36 of the direct multiply-accumulates" if wino else "direct implicit GEMM",
                    "executed_frac": fl / sec / 1e12 / PEAK_FP32_MFMA_TFLOPS / (2.25 if wino else 1.0),
                    "launches_per_step": n // args.profile_steps, "avg_launch_ms": sec / n * 1e3,
                    "avg_launch_gflop": fl / n / 1e9,
                    "by_kernel": {k: {"tflops": v[0] / v[1] / 1e12, "ms_per_step": v[1] / args.profile_steps * 1e3,
                                      "launches_per_step": v[2] // args.profile_steps} for k, v in sorted(agg.items())}}
    elif world > 1:
        pass

    if world > 1:
        dist.barrier()
    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(args.height, args.width, args.cpu_batch, args.cpu_steps, 1)
        total_images = args.batch * world * args.steps
        ips = total_images / dt
        step_flops = TRAIN_GFLOP_PER_IMG * 1e9 * args.batch if (args.height, args.width) == (128, 416) else None
        line = {
            "metric": "images/sec fwd+bwd Disp_vgg_BN 128x416 b32 @1/2/4/8 GPU",
            "value": ips, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Disp_vgg_BN L1-loss training step (fwd + 1/disp + masked L1 + bwd + Adam), synthetic KITTI "
                                   "%dx%d, batch %d per GPU" % (args.height, args.width, args.batch),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world, "final_loss": final_loss},
            "step_tflops_per_gpu": (step_flops / (dt / args.steps) / 1e12) if step_flops else None,
            "step_frac_of_fp32_mfma_peak": (step_flops / (dt / args.steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS) if step_flops else None,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` (read + write) from the committed rocprofv3 PMC passes (tools/pmc_traffic.sh: separate
    FETCH_SIZE / WRITE_SIZE runs of this same command, corrected and calibrated as profiles/*pmc_traffic.json states), or None.
    The counters cannot be collected from inside this process, so the latest committed summary is reported."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))["kernels"].get(kernel)
        return (k["read_bytes"] + k["write_bytes"]) if k else None
    except Exception:
        return None


def _quiet_init(net):
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        net.init_weights(use_pretrained_weights=False)


if __name__ == "__main__":
    main()
