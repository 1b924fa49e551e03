#!/usr/bin/env python3
"""Headline benchmark: images/sec, forward + backward (+ Adam step), Disp_vgg_BN, 128x416, batch 32 per GPU, synthetic
KITTI-like data resident in HBM (BASELINE.json metric / configs[1]).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --gpus N ...          (no launcher: spawns the N ranks itself through torch.distributed.run on a free port)

One rank per GPU (RCCL over xGMI); the only exchange is the bucketed gradient all-reduce overlapped with the encoder backward.
Rank 0 prints ONE JSON line.

  --scaling strong  (default) the literal metric of BASELINE.json ("b32 @1/2/4/8 GPU"): --global-batch 32 split into 32/N images per
                    rank, total work fixed as N grows; `value` is that rate.  At N = 1 this IS b32 on one GPU.  With N > 1 the line also
                    carries `weak_scaling` (every rank stepping its own b32, timed after the headline region) as a secondary figure.
  --scaling weak    every rank steps its own batch of --batch images (b32): per-GPU work fixed as N grows.
  --config          vggbn128 (the headline) | vggbn480 | res50_480 | dorn128 | photo128: BASELINE.json configs[2..4] and the
                    480x640 secondary of SURVEY 8d produce their own lines (never the headline; `metric` says which).

  --compute         f32x3 (default = the library's default) | f32 | bf16: arithmetic of the Winograd forward / input-gradient kernels.
                    f32x3 forms every fp32 product on the bf16 matrix cores from three exact bf16 pieces per operand (six partial
                    products, fp32 accumulation): an fp32 result -- error against fp64 at or below the fp32 instruction's
                    (tests/test_gpu_kernels.py::test_winograd_error_vs_fp64, tools/ubench/bf16x3.hip); every default run also times the
                    same step with the fp32 matrix instruction (`f32_mfma_path`), so both numbers come from one run on one box.

  --launch          auto (default) | tape | eager | graph: how the host issues the step.  tape: the kernel launches and stream fences of ONE
                    real step are recorded inside libdispnet_hip (dn_tape_*, graph.TapedStep) and re-issued by one C call per step -- the
                    same kernels on the same streams as the eager launches, ~2 us of host time each instead of ~20 (at 32 / 8 = 4 images
                    per GPU the Python side of a step costs as much as the device side); under data parallelism the tape is cut at every
                    gradient bucket and the all-reduce is issued live in between.  Before the timed region one replay and one eager step
                    from the same state must agree bit for bit (config.tape_verified), else the run uses eager launches and says so.
                    auto = tape for the metric's config.  graph = hipGraph replay (slower on the device than eager on ROCm 7.2).
  --adam-overlap    auto (default) | 0 | 1: the Adam update of a gradient bucket right behind the bucket's completion / all-reduce, under
                    the rest of the backward pass (bit-identical to one update); auto = above 16 images per GPU.

Besides the contract fields the line carries
  roofline      the dominant matrix-pipe kernel (by time).  `achieved` = multiply-accumulates the kernel EXECUTES on its matrix pipe x 2
                divided by its launch durations (HIP events on the launch stream, instrumented steps after the timed region); `peak` =
                the dense peak of THAT pipe (2500 TFLOP/s bf16 for the f32x3 / bf16 Winograd variants, which execute six / one bf16
                partial products per multiply; 157.3 TFLOP/s for the fp32 matrix instruction); `frac` <= 1 by construction.
                `fp32_equivalent_achieved` counts each fp32 multiply-accumulate once (what a v_mfma_f32 kernel would have to sustain).
                The Winograd F(2x2,3x3) kernels execute 16/36 of the direct convolution's MACs: the direct-FLOP rate BASELINE.md
                section 2 counts is reported separately as `credited_achieved` / `credited_frac`.
  roofline_hbm  the slowest HBM-bound family of the step: algorithmic bytes (operands read once, results written once) over its
                event time, against 8 TB/s and against the float4 copy rate measured on this box (dn_ubench_copy).
  step_executed_frac  all conv-family launches' fp32-equivalent executed FLOPs (each fp32 multiply-accumulate once) over the step time, / 157.3.
  f32_mfma_path the same step timed with --compute f32 semantics (fp32 matrix instruction everywhere) after the headline region.
  cpu_baseline  the CPU oracle (oracle/, PyTorch-CPU restatement pinned to the reference's golden vectors; kind "port")
                running the same training step on this box's host cores per SURVEY 8d: batch 8, 2 warm-up + 6 timed steps at each of 32 / 64 / 128
                torch threads (~20 s of CPU work); `value` is the best of the three, all three are in `sample` / `by_threads`.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

PEAK_BF16_MFMA_TFLOPS = 2500.0       # same guide: dense bf16 matrix peak (the pipe the f32x3 / bf16 Winograd variants run on)
X3_PRODUCTS = 6                      # f32x3: six bf16 partial products per fp32 multiply
PEAK_FP32_MFMA_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_HBM_GBPS = 8000.0               # same guide: HBM3E spec (6.29 TB/s measured float4 copy there)
THIN_KERNELS = ("stem3_conv_kernel", "lds3_conv_kernel", "lds3k_conv_kernel", "lds3_wgrad", "lds3k_wgrad", "stem_conv_kernel", "thin_conv_kernel", "thin_wgrad_kernel", "head_fwd", "head_dgrad", "head_wgrad", "igemm_conv_u32_kernel<128, 32, 32, 32>",
                "igemm_wgrad_kernel<32, 32, 32, true>", "igemm_wgrad_u32_kernel<32, 32, 32, false>")
WINO_EXEC = 16.0 / 36.0              # F(2x2,3x3): 16 element-wise products per 2x2 tile instead of 36 MACs

# name -> (metric text, network, H, W, default batch per GPU, dataset tag, BASELINE.md section 2 train GFLOP/img or None)
CONFIGS = {
    "vggbn128": ("images/sec fwd+bwd Disp_vgg_BN 128x416 b32 @1/2/4/8 GPU", "Disp_vgg_BN", 128, 416, 32, "kitti", 110.54),
    "vggbn480": ("images/sec fwd+bwd Disp_vgg_BN 480x640 b16 (secondary, SURVEY 8d)", "Disp_vgg_BN", 480, 640, 16, "nyu", 637.7),
    "res50_480": ("images/sec fwd+bwd Disp_res_50 480x640 b16 (BASELINE configs[3])", "Disp_res_50", 480, 640, 16, "nyu", 247.3),
    "dorn128": ("images/sec fwd+bwd Disp_vgg_BN_DORN K=80 128x416 b32 (BASELINE configs[4], fp32)", "Disp_vgg_BN_DORN", 128, 416, 32,
                "kitti", 111.3),
    "photo128": ("images/sec fwd+bwd Disp_vgg_BN + PoseExpNet photometric warp loss seq-len 3 128x416 b32 (BASELINE configs[2])",
                 "Disp_vgg_BN", 128, 416, 32, "kitti", None),
}


def synthetic_batch(batch, h, w, device, seed, dataset="kitti"):
    """U(0,1) image normalised (x-0.5)/0.5; kitti: 5 %-dense U(1,80) ground truth; nyu: dense U(0.5,10)."""
    g = torch.Generator().manual_seed(seed)
    img = (torch.rand(batch, 3, h, w, generator=g) - 0.5) / 0.5
    if dataset == "nyu":
        gt = torch.rand(batch, h, w, generator=g) * 9.5 + 0.5
    else:
        depth = torch.rand(batch, h, w, generator=g) * 79.0 + 1.0
        gt = depth * (torch.rand(batch, h, w, generator=g) < 0.05).float()
    return img.to(device), gt.to(device)


def host_description():
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return model, len(phys) or None, os.cpu_count() or 1


def _cpu_oracle_rate(h, w, batch, steps, warmup, threads, cpus=None, seed=0, start_at=None):
    """img/s of the oracle's Disp_vgg_BN + l1_loss + Adam training step with `threads` torch threads (optionally pinned to `cpus`;
    `start_at`: a wall-clock time all replicas of a multi-process sample begin their timed steps at)."""
    if cpus:
        try:
            os.sched_setaffinity(0, cpus)
        except OSError:
            pass
    from oracle import losses as OL, nets as ON
    torch.set_num_threads(threads)
    img, gt = synthetic_batch(batch, h, w, "cpu", seed)
    sd = ON.xavier_init_(ON.disp_vgg_bn_state_dict(), torch.Generator().manual_seed(0))
    params = []
    for k, v in sd.items():
        if torch.is_floating_point(v) and "running" not in k:
            v.requires_grad_(True)
            params.append(v)
    opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.999))
    t0 = None
    for it in range(warmup + steps):
        if it == warmup:
            if start_at is not None:
                time.sleep(max(0.0, start_at - time.time()))
            t0 = time.perf_counter()
        depth = [1 / d for d in ON.disp_vgg_bn(sd, img, training=True)]
        loss = OL.l1_loss(gt, depth, "kitti")
        opt.zero_grad()
        loss.backward()
        opt.step()
    dt = time.perf_counter() - t0
    return batch * steps / dt, dt


def _cpu_replica_main(spec):
    """`python bench.py --cpu-replica JSON`: one pinned replica of the CPU baseline; prints one JSON line."""
    j = json.loads(spec)
    rate, dt = _cpu_oracle_rate(j["h"], j["w"], j["batch"], j["steps"], j["warmup"], j["threads"], set(j["cpus"]), j["seed"], j["start_at"])
    print(json.dumps({"rate": rate, "dt": dt}), flush=True)


def cpu_baseline(h, w, batch, steps, warmup, threads=None):
    """The oracle's Disp_vgg_BN + l1_loss + Adam training step on the host cores (b8, warm-up + timed steps; kind "port").
    Two layouts are timed and the better one is `value`:
      * one process at 32 / 64 / 128 torch threads (DN_CPU_THREADS=n[,n..] overrides): a quick look at every count with the SAME
        short budget (1 warm-up + 2 steps), then the winner re-measured with the full sample;
      * data-parallel replicas -- what the GPU side does, and what a 2-socket host needs to use its cores at this problem size (one
        process's oneDNN threads stop scaling at ~32; round 1 measured 0.04 img/s with every logical CPU in one pool): P processes of
        the winning thread count, each pinned to its own block of physical cores, each stepping its own batch, started together;
        aggregate img/s = P * batch * steps / the slowest replica's time (DN_CPU_REPLICAS=0 skips it).
    `cores` = the threads the reported layout actually used."""
    model, physical, logical = host_description()
    if threads:
        sweep = [int(threads)]
    elif os.environ.get("DN_CPU_THREADS"):
        sweep = [int(t) for t in os.environ["DN_CPU_THREADS"].split(",")]
    else:
        sweep = sorted({min(t, logical) for t in (32, 64, 128)})
    quick = {}
    for cores in sweep:
        quick[cores] = _cpu_oracle_rate(h, w, batch, 2, 1, cores)[0] if len(sweep) > 1 else 0.0
    best = max(quick, key=quick.get)
    single = _cpu_oracle_rate(h, w, batch, steps, warmup, best)[0]
    layouts = {"1x%d" % best: single}
    value, used = single, best
    want = os.environ.get("DN_CPU_REPLICAS")
    phys = physical or logical // 2 or 1
    nrep = int(want) if want else max(1, min(phys // best, 8))
    if nrep > 1:
        import subprocess
        procs = []
        try:
            try:
                avail = sorted(os.sched_getaffinity(0))
            except AttributeError:
                avail = list(range(logical))
            per = max(1, len(avail) // nrep)
            rsteps = max(2, steps // 2)
            start_at = time.time() + 30.0                          # (interpreter + import torch + one warm-up step of every replica)
            for r in range(nrep):
                spec = {"h": h, "w": w, "batch": batch, "steps": rsteps, "warmup": 1, "threads": min(best, per),
                        "cpus": avail[r * per:(r + 1) * per], "seed": r, "start_at": start_at}
                procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-replica", json.dumps(spec)],
                                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
            budget = 45.0 + 4.0 * rsteps * batch / max(single, 0.05)   # start delay + several times the single-process time of the sample
            out = []
            for pr in procs:
                o, _ = pr.communicate(timeout=max(5.0, start_at + budget - time.time()))
                out.append(json.loads(o.strip().splitlines()[-1]))
            slowest = max(o["dt"] for o in out)
            agg = nrep * batch * rsteps / slowest
            layouts["%dx%d" % (nrep, min(best, per))] = agg
            if agg > value:
                value, used = agg, nrep * min(best, per)
        except Exception as e:                                     # noqa: BLE001 -- the baseline must never take the bench line down
            layouts["replicas_error"] = "%s: %s" % (type(e).__name__, str(e)[:120])
        finally:
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()
    return {"value": value, "unit": "images/sec", "cores": used, "kind": "port",
            "host_cpu": model, "host_physical_cores": physical, "host_logical_cpus": logical,
            "by_threads_quick": {str(k): v for k, v in quick.items()}, "by_layout": layouts,
            "sample": "oracle Disp_vgg_BN+L1+Adam, %dx%d, batch %d per process; one process: 1 warm-up + 2 steps at each of %s torch threads, the "
                      "winner (%d) re-measured with %d warm-up + %d timed steps = %.2f img/s; replicas (processes x threads, pinned, started "
                      "together, %d timed steps each): %s; host: %s physical cores / %d logical CPUs" % (
                          h, w, batch, sorted(quick), best, warmup, steps, single, max(2, steps // 2),
                          ", ".join("%s: %s" % (k, ("%.2f" % v) if isinstance(v, float) else v) for k, v in layouts.items()), physical, logical)}


def measured_peaks(dev):
    """Attainable peaks on THIS box (SURVEY 8d "Peaks"): float4 streaming copy GB/s and register-resident fp32 MFMA TFLOP/s."""
    from supervised_dispnet_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    n = 1 << 28                                           # 1 GiB in, 1 GiB out: far beyond the 256 MiB infinity cache
    src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    best_copy = 0.0
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call("dn_ubench_copy", src.data_ptr(), dst.data_ptr(), n, st)
        e1.record()
        e1.synchronize()
        best_copy = max(best_copy, 8.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del src, dst
    blocks, iters = 256 * 8, 2000
    out = torch.empty(blocks * 256, dtype=torch.float32, device=dev)
    best_mfma = 0.0
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call("dn_ubench_mfma_f32", out.data_ptr(), blocks, iters, st)
        e1.record()
        e1.synchronize()
        best_mfma = max(best_mfma, lib.dn_ubench_mfma_f32_flops(blocks, iters) / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    return {"copy_GBps": best_copy, "mfma_f32_TFLOPs": best_mfma}


def build_workload(cfg, batch, dev, seed, models, LF, U, reciprocal, FusedAdam):
    """-> (step callable, optimizer, description).  Every config: forward -> loss -> zero_grad/backward -> (all-reduce) -> Adam."""
    from supervised_dispnet_amd.graph import backward as seeded_backward
    metric, netname, H, W, _b, ds, _gf = CONFIGS[cfg]
    torch.manual_seed(0)                                   # identical random-init replicas on every rank
    if netname == "Disp_vgg_BN":
        net = models.Disp_vgg_BN(datasets=ds, with_classifier=False)
    elif netname == "Disp_vgg_BN_DORN":
        net = models.Disp_vgg_BN_DORN(datasets=ds, ordinal_c=80, with_classifier=False)
    else:
        net = models.Disp_res_50(datasets=ds)
    _quiet_init(net)
    net.to(dev).train()
    params = list(net._hot_parameters())
    order = list(net._grad_production_order())
    pose_net = None
    if cfg == "photo128":
        pose_net = models.PoseExpNet(nb_ref_imgs=2, output_exp=False).to(dev)
        pose_net.init_weights()
        pose_net.train()
        params += list(pose_net._hot_parameters())
    opt = FusedAdam(params, lr=1e-4, betas=(0.9, 0.999), production_order=order)
    img, gt = synthetic_batch(batch, H, W, dev, seed, ds)
    state = {"reducer": None, "nets": [n_ for n_ in (net, pose_net) if n_ is not None]}

    def finish_step(loss):
        opt.zero_grad()
        seeded_backward(loss)                # loss.backward() from a persistent ones tensor (a launch tape cannot see autograd's fill)
        red = state["reducer"]
        opt.step(grad_scale=red.finish() if red is not None else 1.0)
        return loss

    if cfg in ("vggbn128", "vggbn480", "res50_480"):
        desc = "%s L1-loss training step (fwd + 1/disp + masked L1 + bwd + Adam), synthetic %s %dx%d" % (netname, ds.upper(), H, W)

        def step():
            depth = [reciprocal(d) for d in net(img)]
            return finish_step(LF.l1_loss(gt, depth, ds))               # README recipe: --loss L1 -s 0
    elif cfg == "dorn128":
        desc = "Disp_vgg_BN_DORN (ordinal_c 80) DORN-loss training step (fwd + SID labels + ordinal loss + bwd + Adam), fp32, synthetic KITTI %dx%d" % (H, W)

        def step():
            target = U.get_labels_sid(gt, ordinal_c=80, dataset=ds)     # train.py:431
            _dec, ordc = net(img)
            return finish_step(LF.DORN_loss(gt, ordc, target, ds))
    else:
        desc = ("Disp_vgg_BN + PoseExpNet (trained) photometric_reconstruction_loss over 2 refs x 4 scales + 0.1 * smooth_loss, "
                "seq-len 3, synthetic KITTI %dx%d" % (H, W))
        g = torch.Generator().manual_seed(seed + 77)
        refs = [(img.cpu() + 0.05 * torch.randn(img.shape, generator=g)).clamp(-1, 1).to(dev) for _ in range(2)]
        K = torch.tensor([[241.67, 0, 204.17], [0, 246.28, 59.0], [0, 0, 1]], dtype=torch.float32)
        Kb = K.repeat(batch, 1, 1).to(dev)
        Kib = torch.inverse(K).repeat(batch, 1, 1).to(dev)

        def step():
            mask, pose = pose_net(img, refs)
            depth = [reciprocal(d) for d in net(img)]
            l1 = LF.photometric_reconstruction_loss(img, refs, Kb, Kib, depth, mask, pose, "euler", "zeros")
            return finish_step(l1 + 0.1 * LF.smooth_loss(depth))
    return step, opt, state, desc


def tape_host_profile(taped, dev):
    """Where the host spends a replay (stderr): per-op host time of three replays, each started on an idle device."""
    import ctypes as C
    from supervised_dispnet_amd import _lib
    lib = _lib.load()
    cap = int(taped.launches + taped.fences + 16)
    ns, names = (C.c_int64 * cap)(), (C.c_char_p * cap)()
    for rep in range(3):
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(taped._stream):
            n = lib.dn_tape_replay_timed(taped.tape, ns, names, cap)
        torch.cuda.synchronize(dev)
        if n < 0:
            print("[tape-host-profile] replay failed: %s" % _lib.last_error(), file=sys.stderr)
            return
        rows = [(ns[i] / 1e3, i, (names[i] or b"?").decode()[:70]) for i in range(min(n, cap))]
        tot = sum(r[0] for r in rows)
        fences = [r for r in rows if r[2] == "fence"]
        cum, marks = 0.0, []
        for r in rows:
            cum += r[0]
            if r[1] % 20 == 0:
                marks.append("#%d@%.0f" % (r[1], cum))
        print("[tape-host-profile] replay %d: %d ops, host %.1f us (launches %.1f us, %d fences %.1f us); cumulative us at op: %s"
              % (rep, n, tot, tot - sum(r[0] for r in fences), len(fences), sum(r[0] for r in fences), " ".join(marks)), file=sys.stderr)
        for us, i, nm in sorted(rows, reverse=True)[:12]:
            print("[tape-host-profile]    op %3d %8.1f us  %s" % (i, us, nm), file=sys.stderr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="vggbn128", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="strong (default): --global-batch images per step over ALL ranks = BASELINE.json's 'b32 @1/2/4/8 GPU'; "
                         "weak: --batch images per rank")
    ap.add_argument("--global-batch", type=int, default=32, help="--scaling strong: images per step over ALL ranks")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (default: the config's; strong scaling: global/N)")
    ap.add_argument("--profile-steps", type=int, default=2, help="instrumented steps (after the timed region) for the roofline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-layer", action="store_true", help="print the per-layer launch table to stderr")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=6)
    ap.add_argument("--cpu-warmup", type=int, default=2)
    ap.add_argument("--compute", default="f32x3", choices=["f32", "f32x3", "bf16"],
                    help="arithmetic of the Winograd forward / input-gradient kernels (tensors and every other kernel are fp32 in all modes). "
                         "f32x3 (default, the library's default): fp32 products formed on the bf16 matrix cores from three exact bf16 pieces "
                         "per operand, fp32 accumulation -- error vs fp64 at or below the fp32 instruction's (tests/test_gpu_kernels.py); "
                         "f32: the fp32 matrix instruction (also timed in every default run: field f32_mfma_path); "
                         "bf16: operands ROUNDED to bf16 (BASELINE configs[4]'s mixed precision; never the headline)")
    ap.add_argument("--alt-steps", type=int, default=10, help="steps of the secondary timings (fp32 matrix instruction; weak scaling at N > 1) (0: skip)")
    ap.add_argument("--rccl-selfcheck", default="auto", choices=["auto", "0", "1"],
                    help="N > 1: after the timed region, all-reduce one 20 MB bucket through this library's own RCCL communicator AND through "
                         "torch.distributed and compare bitwise (config.rccl_selfcheck); guarded by a watchdog that prints the line anyway")
    ap.add_argument("--graph", default="0", choices=["0", "1"],
                    help="1: replay the whole step (fwd + loss + bwd + Adam) as ONE captured hipGraph.  Off by default: measured on "
                         "ROCm 7.2 the replay is 1-4 %% SLOWER than the eager launches at every batch size (profiles/r02_strong_1gpu.txt)")
    ap.add_argument("--launch", default="auto", choices=["auto", "eager", "graph", "tape"],
                    help="how the host issues the step.  tape: the launches of one real step are recorded once by libdispnet_hip (dn_tape_*) "
                         "and re-issued by one C call per step -- same kernels, streams and fences as eager, ~2 us of host time per launch "
                         "instead of ~20 (the step is launch-bound at 4 images per GPU); graph: hipGraph replay (= --graph 1); "
                         "auto: tape for the metric's config when the recording succeeds, else eager (config.launch says which ran)")
    ap.add_argument("--tape-verify", default="1", choices=["0", "1"],
                    help="--launch tape: before the timed region, one replay and one eager step from the same state must agree bit for bit "
                         "(parameters, Adam moments, counters, BatchNorm buffers, loss): config.tape_verified")
    ap.add_argument("--adam-overlap", default="auto", choices=["auto", "0", "1"],
                    help="1: the Adam update of a gradient bucket runs as soon as the bucket (and its all-reduce) is complete, on its own "
                         "stream under the rest of the backward pass (FusedAdam.overlap_backward; bit-identical to the single update); "
                         "auto: on for the metric's config")
    ap.add_argument("--reducer", default="auto", choices=["auto", "rccl1"],
                    help="one rank only.  rccl1: run the step with the data-parallel machinery live -- gradient buckets, the launch tape cut "
                         "at every bucket, event fences against both compute streams, ncclAllReduce on this library's own single-rank RCCL "
                         "communicator and stream, 1/N folded into Adam -- so that the host-side cost of the N > 1 path is a number on the "
                         "one-GPU pool (only the wire is missing); config.comm says what ran")
    ap.add_argument("--comm-standin", type=int, default=0, metavar="K",
                    help="with --reducer rccl1: after each bucket's all-reduce, K round trips of the bucket through a scratch buffer on the "
                         "communication stream (dn_ubench_copy: read + write of the bucket twice per trip) -- a stand-in for the HBM side of "
                         "a ring all-reduce competing with the backward pass.  A PROJECTION, labelled as such in config.comm_standin")
    ap.add_argument("--wgrad-streams", type=int, default=0, help="A/B: side streams the weight gradients alternate between (0: the engine's rule)")
    ap.add_argument("--no-fold", action="store_true",
                    help="A/B: run dn_bn_finalize / the BatchNorm-backward sums as launches of their own instead of in the last-arriving block "
                         "of the Winograd kernels (engine.FOLD_FINALIZE; same bits either way)")
    ap.add_argument("--system-fences", action="store_true",
                    help="A/B: the launch tape's fences between the main and the weight-gradient streams with the system-scope release HIP events "
                         "carry by default (engine.DEVICE_SCOPE_FENCES = False)")
    ap.add_argument("--no-loss-fusion", action="store_true",
                    help="A/B: the masked loss forward as three launches + a fill (loss_functions.FUSE_MASKED_FWD = False) and an owned copy of the "
                         "gradient autograd seeds the finest head with (engine.BORROW_SEED_GRADS = False)")
    ap.add_argument("--tape-join-every-step", action="store_true",
                    help="A/B: every replay makes the caller's stream wait for the tape's (graph.TapedStep's default; bench.py reads nothing "
                         "between steps and joins once after the timed loop)")
    ap.add_argument("--tape-host-profile", action="store_true",
                    help="after the timed region: host time of every launch / fence of three replays (dn_tape_replay_timed), the slowest to stderr")
    ap.add_argument("--watchdog-s", type=float, default=0.0,
                    help="N > 1: if the line has not been printed after this many seconds (default 900 at N > 1, 0 = off at N = 1) rank 0 "
                         "prints a line with value null and the reason, and every rank exits non-zero")
    ap.add_argument("--extras", default="auto", choices=["auto", "0", "1"],
                    help="N = 1, default workload: after the headline's timed region, also time BASELINE configs[2..4] + the 480x640 VGG step "
                         "(10 steps each) and the 4 / 8 / 16-image steps of the metric's model (launch tape; + the data-parallel machinery on a "
                         "single-rank own-RCCL communicator), each as a nested run of this script, and carry them in the SAME line as "
                         "`other_configs` / `strong_1gpu` (auto: on for the plain `python bench.py [--gpus 1]` call)")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU-only plumbing check of the N>1 path: build the net, the arena, the buckets, one fake all-reduce cycle, tear down")
    args = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # invoked plainly (`python bench.py --gpus N`, as the reference's train.py needs no launcher either: train.py:316-317):
        # become the launcher -- one rank per GPU under torch.distributed.run on a free local port; rank 0 prints the line
        return self_launch(args.gpus)
    if args.dry_run:
        return dry_run(args, world, rank)
    watchdog = start_watchdog(args, world, rank)
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
    import supervised_dispnet_amd.utils as U
    from supervised_dispnet_amd import engine
    from supervised_dispnet_amd.distributed import GradReducer
    from supervised_dispnet_amd.functional import reciprocal
    from supervised_dispnet_amd.optim import FusedAdam

    engine.set_compute(args.compute)
    if args.system_fences:
        engine.DEVICE_SCOPE_FENCES = False
    if args.no_loss_fusion:
        import supervised_dispnet_amd.loss_functions as _LF
        _LF.FUSE_MASKED_FWD = False
        engine.BORROW_SEED_GRADS = False
    if args.no_fold:
        engine.FOLD_FINALIZE = False
    if args.wgrad_streams > 0:
        engine.WGRAD_STREAMS = args.wgrad_streams
    global PMC_CONFIG
    PMC_CONFIG = args.config
    metric, netname, H, W, cfg_batch, ds, gflop_img = CONFIGS[args.config]
    if args.compute == "bf16":
        metric = metric.replace(", fp32)", ", mixed precision: bf16 multiplies / fp32 accumulation in the Winograd kernels)")
    if args.scaling == "strong":
        gb = args.global_batch if args.global_batch > 0 else cfg_batch
        if args.config != "vggbn128" and args.global_batch == 32:
            gb = cfg_batch                               # the other configs are quoted on their own batch (b16 at 480x640)
        if gb % world:
            raise SystemExit("global batch %d does not divide over %d ranks" % (gb, world))
        batch = args.batch or gb // world
    else:
        batch = args.batch or cfg_batch
    step, opt, state, desc = build_workload(args.config, batch, dev, rank, models, LF, U, reciprocal, FusedAdam)
    # (measured on one box, tape launches: b32 17.83 -> 17.72 ms with it; b8 5.83 -> 5.88 and b4 4.00 -> 4.04 -- at small batches the
    #  weight-gradient side streams the ranges run on are themselves close to critical -- so "auto" follows the per-GPU batch)
    overlap_adam = args.adam_overlap == "1" or (args.adam_overlap == "auto" and args.config == "vggbn128" and batch * H * W > 16 * 128 * 416)
    # one rank: the reducer exchanges nothing (comm "torch" at world 1 = no collective) and only tells the optimizer when a bucket of
    # gradients is complete
    standin = None
    if args.reducer == "rccl1" and world == 1:
        reducer = GradReducer(opt.arena, comm="rccl")
        if reducer.comm is None:
            raise SystemExit("--reducer rccl1: the single-rank RCCL communicator could not be created")
        if args.comm_standin > 0:
            standin = {"round_trips_per_bucket": args.comm_standin, "buckets": len(reducer.buckets),
                       "bytes_moved_per_step": sum((b["hi"] - b["lo"]) * 4 for b in reducer.buckets) * 4 * args.comm_standin,
                       "note": "PROJECTION: dn_ubench_copy of every gradient bucket to a scratch buffer and back on the communication stream "
                               "behind its single-rank ncclAllReduce -- the HBM side of a ring all-reduce under the backward pass; no xGMI "
                               "link is involved"}
            reducer.comm.standin_round_trips = args.comm_standin
    else:
        reducer = GradReducer(opt.arena, comm=(None if world > 1 else "torch")) if (world > 1 or overlap_adam) else None
    if overlap_adam:
        opt.overlap_backward(reducer)
    state["reducer"] = reducer
    engine.GradSink.reducer = reducer

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    eager_step, graphed, graph_note = step, False, None
    launch_mode = "graph" if args.graph == "1" else args.launch
    if launch_mode == "auto":
        launch_mode = "tape" if args.config == "vggbn128" else "eager"
    taped, tape_verified = None, None
    if launch_mode == "tape":
        from supervised_dispnet_amd.graph import TapedStep
        for _ in range(2):
            eager_step()
        try:
            taped = TapedStep(eager_step, optimizer=opt, warmup=2, lazy_join=not args.tape_join_every_step).capture()
            step = taped
            if args.tape_verify == "1":
                # one replay against one eager step from the same parameters / moments / counters / BatchNorm buffers, bit for bit
                st = [opt.arena.flat_p, opt.exp_avg, opt.exp_avg_sq, opt._dev["step"], opt._dev["derived"]]
                st += [b for n_ in state.get("nets", ()) for b in n_.buffers() if b.is_cuda]
                tape_verified = taped.verify(st)
                if not tape_verified[0]:               # never time a replay that is not the eager step
                    graph_note = "launch tape refused: replay differs from the eager step (max |diff| %.3g)" % tape_verified[1]
                    step, taped = eager_step, None
        except Exception as e:                        # noqa: BLE001 -- the eager path is the same kernels; say why it was used
            graph_note = "%s: %s" % (type(e).__name__, str(e)[:200])
            taped = None
            torch.cuda.synchronize()
    if launch_mode == "graph":
        from supervised_dispnet_amd.graph import GraphedStep
        for _ in range(2):
            eager_step()                              # the arena / caches exist before the capture
        try:
            gs = GraphedStep(eager_step, optimizer=opt, warmup=2).capture()
            step, graphed = gs, True
        except Exception as e:                        # noqa: BLE001 -- the eager path is the same kernels; say why it was used
            graph_note = "%s: %s" % (type(e).__name__, str(e)[:200])
            opt.capturable(False)
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    import gc
    gc.collect()
    gc.disable()                 # a generational collection inside the timed loop stalls the launch thread for milliseconds
    fence()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    mark_stream = taped._stream if (taped is not None and taped.lazy_join) else torch.cuda.current_stream()     # where the step's work is
    if mark_stream != torch.cuda.current_stream():
        mark_stream.wait_stream(torch.cuda.current_stream())
    marks[0].record(mark_stream)
    for i in range(args.steps):
        loss = step()
        marks[i + 1].record(mark_stream)
    if taped is not None:
        taped.join()
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    per_step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())

    # ---- secondary timing (not `value`): the same step with the fp32 matrix instruction in the Winograd kernels, so that the line
    #      carries both numbers from one run on one box
    alt = None
    if args.compute == "f32x3" and args.alt_steps > 0 and not graphed:
        engine.set_compute("f32")
        for _ in range(2):
            eager_step()
        gc.collect()
        gc.disable()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.alt_steps):
            eager_step()
        fence()
        adt = time.perf_counter() - t0
        gc.enable()
        if world > 1:
            t = torch.tensor([adt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            adt = float(t.item())
        alt = {"compute": "f32 (v_mfma_f32_32x32x2_f32 in every matrix kernel)", "value": batch * world * args.alt_steps / adt, "unit": "images/sec",
               "ms_per_step": adt / args.alt_steps * 1e3, "steps": args.alt_steps}
        engine.set_compute(args.compute)
        for _ in range(2):
            eager_step()
        torch.cuda.synchronize()

    # ---- secondary timing at N > 1 (not `value`): weak scaling, every rank stepping its own full batch of the config
    weak = None
    if world > 1 and args.scaling == "strong" and args.alt_steps > 0 and batch != cfg_batch:
        engine.GradSink.reducer = None
        wstep, wopt, wstate, _wdesc = build_workload(args.config, cfg_batch, dev, rank, models, LF, U, reciprocal, FusedAdam)
        wred = GradReducer(wopt.arena)
        wstate["reducer"] = wred
        engine.GradSink.reducer = wred
        for _ in range(3):
            wstep()
        gc.collect()
        gc.disable()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.alt_steps):
            wstep()
        fence()
        wdt = time.perf_counter() - t0
        gc.enable()
        t = torch.tensor([wdt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wdt = float(t.item())
        weak = {"scaling": "weak", "batch_per_gpu": cfg_batch, "global_batch": cfg_batch * world, "value": cfg_batch * world * args.alt_steps / wdt,
                "unit": "images/sec", "ms_per_step": wdt / args.alt_steps * 1e3, "steps": args.alt_steps, "comm": wred.path}
        engine.GradSink.reducer = reducer
        del wstep, wopt, wstate, wred
        torch.cuda.synchronize()

    if args.tape_host_profile and taped is not None and rank == 0:
        tape_host_profile(taped, dev)

    # ---- instrumented steps (not part of `value`): HIP events around every conv-family launch and every HBM-bound family
    roofline = roofline_hbm = None
    step_exec_flops = step_credited_flops = None
    step = eager_step                            # per-launch event timing needs the launches issued one by one
    if args.profile_steps > 0 and rank != 0:
        for _ in range(args.profile_steps):      # the steps carry collectives: every rank takes them, rank 0 records
            step()
        torch.cuda.synchronize()
    if rank == 0 and args.profile_steps > 0:
        peaks = measured_peaks(dev)
        engine.PROFILE = []
        for _ in range(args.profile_steps):
            step()
        torch.cuda.synchronize()
        prof, engine.PROFILE = engine.PROFILE, None
        nps = args.profile_steps
        if args.per_layer:
            seen = {}
            for name, flops, e0, e1, tag, nbytes, _ab in prof:
                r = seen.setdefault((name, tag), [0.0, 0.0, 0, 0])
                r[0] += flops; r[1] += e0.elapsed_time(e1) * 1e-3; r[2] += 1; r[3] += nbytes
            print("%-46s %-58s %9s %9s %8s %8s %6s %8s" % ("kernel", "layer / entry", "ms/launch", "GFLOP", "TFLOP/s", "GB/s", "n/step", "ms/step"), file=sys.stderr)
            for (name, tag), (fl, sec, n, nb) in seen.items():
                print("%-46s %-58s %9.3f %9.2f %8.1f %8.0f %6.1f %8.3f" % (name, tag, sec / n * 1e3, fl / n / 1e9, fl / sec / 1e12, nb / sec / 1e9, n / nps, sec / nps * 1e3), file=sys.stderr)
        mf, hb, thin = {}, {}, {}
        for name, flops, e0, e1, _tag, nbytes, abytes in prof:
            sec = e0.elapsed_time(e1) * 1e-3
            if any(t in name for t in THIN_KERNELS):      # the full-resolution thin layers: both rooflines, per kernel
                a = thin.setdefault(name, [0.0, 0.0, 0, 0])
                a[0] += flops; a[1] += sec; a[2] += 1; a[3] += abytes
            if nbytes:
                a = hb.setdefault(name, [0, 0.0, 0])
                a[0] += nbytes; a[1] += sec; a[2] += 1
            else:
                a = mf.setdefault(name, [0.0, 0.0, 0])
                a[0] += flops; a[1] += sec; a[2] += 1
        # multiply-accumulates x 2 the algorithm needs in fp32 terms (Winograd: 16/36 of the direct ones) ...
        execf = lambda k, fl: fl * (WINO_EXEC if "wino_" in k else 1.0)
        # ... and what the matrix pipe the kernel runs on executes for them (f32x3: six bf16 partial products per fp32 multiply)
        is_x3 = lambda k: ("wino_conv_kernel" in k and k.endswith(", 3>")) or "wino_conv8_kernel" in k or "wino_wgrad_x3" in k
        is_bf = lambda k: "wino_conv_kernel" in k and k.endswith(", 1>")
        pipef = lambda k, fl: execf(k, fl) * (X3_PRODUCTS if is_x3(k) else 1.0)
        step_credited_flops = sum(v[0] for v in mf.values()) / nps
        step_exec_flops = sum(execf(k, v[0]) for k, v in mf.items()) / nps
        name, (fl, sec, n) = max(mf.items(), key=lambda kv: kv[1][1])
        wino = "wino_" in name
        ach = pipef(name, fl) / sec / 1e12
        on_bf16_pipe = is_x3(name) or is_bf(name)
        kpeak = PEAK_BF16_MFMA_TFLOPS if on_bf16_pipe else PEAK_FP32_MFMA_TFLOPS
        algo = "direct implicit GEMM"
        if wino:
            algo = "winograd F(2x2,3x3): executes 16/36 of the direct multiply-accumulates"
            if is_x3(name):
                algo += "; each fp32 product = six bf16 partial products (three exact bf16 pieces per operand) on v_mfma_f32_32x32x16_bf16, fp32 accumulation"
        roofline = {"bound": "mfma", "kernel": name, "achieved": ach, "peak": kpeak, "unit": "TFLOP/s",
                    "frac": ach / kpeak, "traffic": pmc_traffic(name), "traffic_source": pmc_traffic_source(name),
                    "pipe": "bf16 matrix cores (dense peak 2500 TFLOP/s)" if on_bf16_pipe else "fp32 matrix instruction (157.3 TFLOP/s)",
                    "fp32_equivalent_achieved": execf(name, fl) / sec / 1e12,
                    "fp32_equivalent_frac_of_fp32_mfma_peak": execf(name, fl) / sec / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                    "limiter": ("L2 -> CU bandwidth and vector-memory issue (ablations DN_WINO_DBG=32/64, DESIGN.md section 3), not the matrix pipe"
                                if is_x3(name) else None),
                    "frac_of_measured_peak": (ach / peaks["mfma_f32_TFLOPs"]) if not on_bf16_pipe else None, "measured_peak": peaks["mfma_f32_TFLOPs"],
                    "algorithm": algo,
                    "credited_achieved": fl / sec / 1e12, "credited_frac": fl / sec / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                    "launches_per_step": n // nps, "avg_launch_ms": sec / n * 1e3, "avg_launch_gflop_executed": execf(name, fl) / n / 1e9,
                    "avg_launch_gflop_credited": fl / n / 1e9,
                    "by_kernel": {k: {"tflops_executed": pipef(k, v[0]) / v[1] / 1e12, "tflops_fp32_equivalent": execf(k, v[0]) / v[1] / 1e12,
                                      "tflops_credited": v[0] / v[1] / 1e12,
                                      "pipe_frac": pipef(k, v[0]) / v[1] / 1e12 / (PEAK_BF16_MFMA_TFLOPS if (is_x3(k) or is_bf(k)) else PEAK_FP32_MFMA_TFLOPS),
                                      "ms_per_step": v[1] / nps * 1e3, "launches_per_step": v[2] // nps} for k, v in sorted(mf.items())}}
        if hb:
            name, (nb, sec, n) = max(hb.items(), key=lambda kv: kv[1][1])
            gbps = nb / sec / 1e9
            roofline_hbm = {"bound": "hbm", "kernel": name, "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                            "frac": gbps / PEAK_HBM_GBPS, "traffic": pmc_traffic(name), "traffic_source": pmc_traffic_source(name), "frac_of_measured_peak": gbps / peaks["copy_GBps"],
                            "measured_peak": peaks["copy_GBps"], "launches_per_step": n // nps, "avg_launch_ms": sec / n * 1e3,
                            "avg_launch_algorithmic_bytes": nb / n,
                            "by_kernel": {k: {"GBps": v[0] / v[1] / 1e9, "ms_per_step": v[1] / nps * 1e3, "launches_per_step": v[2] // nps}
                                          for k, v in sorted(hb.items())},
                            # the thin full-resolution convolutions (3 / 16 / 17 / 32 channels at 128x416 and 64x208, one-channel heads):
                            # algorithmic bytes AND multiply-accumulates of each against BOTH ceilings -- at 16 channels x 9 taps their
                            # arithmetic intensity (~38 flop/B) is above the fp32 matrix instruction's balance point (157.3 TFLOP/s /
                            # 8 TB/s = 20 flop/B), so that instruction, not HBM, is what bounds them on this path
                            "thin_layers": {k: {"GBps": v[3] / v[1] / 1e9, "frac_of_hbm_peak": v[3] / v[1] / 1e9 / PEAK_HBM_GBPS,
                                                "tflops": v[0] / v[1] / 1e12, "frac_of_fp32_mfma_peak": v[0] / v[1] / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                                                "ms_per_step": v[1] / nps * 1e3, "launches_per_step": v[2] // nps,
                                                "hbm_bound_ms_per_step": v[3] / nps / (PEAK_HBM_GBPS * 1e9) * 1e3,
                                                "mfma_bound_ms_per_step": v[0] / nps / (PEAK_FP32_MFMA_TFLOPS * 1e12) * 1e3,
                                                "frac_of_binding_roofline": max(v[3] / (PEAK_HBM_GBPS * 1e9), v[0] / (PEAK_FP32_MFMA_TFLOPS * 1e12)) / v[1]}
                                            for k, v in sorted(thin.items())}}

    if world > 1:
        dist.barrier()
    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1 and args.config == "vggbn128":
            cpu = cpu_baseline(H, W, args.cpu_batch, args.cpu_steps, args.cpu_warmup)
        total_images = batch * world * args.steps
        sec_step = dt / args.steps
        line = {
            "metric": metric, "value": total_images / dt, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": sec_step * 1e3, "ms_per_step_median": statistics.median(per_step_ms), "ms_per_step_min": min(per_step_ms),
            "ms_per_step_max": max(per_step_ms), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None,
            "dtype": {"f32": "f32",
                      "f32x3": "f32 (Winograd forward / input gradient / weight gradient form each f32 product from three exact bf16 pieces per operand on the bf16 matrix cores, six partial products, f32 accumulate; error vs fp64 at or below the f32 matrix instruction's)",
                      "bf16": "bf16 multiply / f32 accumulate in the Winograd forward + input gradient; f32 tensors, f32 everywhere else"}[args.compute],
            "data": "synthetic",
            "config": {"workload": "%s, batch %d per GPU" % (desc, batch), "name": args.config, "global_batch": batch * world,
                       "parallelism": "dp%d" % world, "final_loss": final_loss,
                       "launch": ("one hipGraph replay per step" if graphed else
                                  ("launch tape: %d launches + %d stream fences (%d as the stop event of the launch in front) of one recorded step re-issued by dn_tape_replay, %d segment(s)"
                                   % (taped.launches, taped.fences, taped.riding_fences, taped.segments)) if taped is not None else "eager launches"),
                       "tape_replay_host_ms": (round(taped.host_s / max(taped.replays, 1) * 1e3, 4) if taped is not None else None),
                       "tape_join": (None if taped is None else
                                     ("lazy: the replays follow each other on the tape's stream, one join after the timed steps (train.py joins and reads "
                                      "the loss every step: about +25 us per step; --tape-join-every-step measures that)" if taped.lazy_join
                                      else "the caller's stream joins the tape's after every replay")),
                       "tape_verified": (None if tape_verified is None else
                                         ("replay == eager step, bit for bit" if tape_verified[0] else "MISMATCH: max |diff| %.3g" % tape_verified[1])),
                       "graph_fallback": graph_note, "adam": "per bucket, under the backward pass" if overlap_adam else "one pass after the backward",
                       "comm": (reducer.path + (" (single-rank communicator: buckets, tape cuts, fences and ncclAllReduce live; no wire)"
                                                if (world == 1 and reducer.comm is not None) else "")) if reducer is not None else "none (one rank)",
                       "comm_buckets": len(reducer.buckets) if reducer is not None else 0,
                       "comm_standin": standin,
                       "rccl_selfcheck": {"status": "not run: " + ("one rank" if world == 1 else ("--rccl-selfcheck 0" if args.rccl_selfcheck == "0" else
                                                                                                 "backend %s" % os.environ.get("DN_DIST_BACKEND", "nccl")))},
                       "dist_backend": dist.get_backend() if world > 1 else None},
            "step_tflops_credited_per_gpu": (step_credited_flops / sec_step / 1e12) if step_credited_flops else None,
            "step_credited_frac": (step_credited_flops / sec_step / 1e12 / PEAK_FP32_MFMA_TFLOPS) if step_credited_flops else None,
            "step_executed_frac": (step_exec_flops / sec_step / 1e12 / PEAK_FP32_MFMA_TFLOPS) if step_exec_flops else None,
            "baseline_md_gflop_per_img": gflop_img,
            "roofline": roofline, "roofline_hbm": roofline_hbm, "f32_mfma_path": alt, "weak_scaling": weak, "cpu_baseline": cpu,
        }
    else:
        line = None
    if world > 1 and args.rccl_selfcheck != "0" and os.environ.get("DN_DIST_BACKEND", "nccl") == "nccl":
        rccl_selfcheck(line, dev, rank, world)
    plain_call = (world == 1 and args.config == "vggbn128" and args.batch == 0 and args.launch == "auto" and args.reducer == "auto" and
                  args.compute == "f32x3" and not args.no_cpu_baseline and "DN_BENCH_NESTED" not in os.environ)
    if line is not None and (args.extras == "1" or (args.extras == "auto" and plain_call)):
        line["other_configs"], line["strong_1gpu"] = run_extras(args)
    if watchdog is not None:
        watchdog.cancel()
    try:
        # librccl prints its version banner through C stdio (buffered when stdout is a pipe): push it out BEFORE the line, so that the
        # JSON line is the last thing this process writes
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:                                           # noqa: BLE001
        pass
    if rank == 0:
        print(json.dumps(line))
        sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()


def _nested(flags, timeout_s=420):
    """One nested run of this script on the same GPU (its own process: its own allocator, tape and knobs); returns its JSON line."""
    import subprocess
    env = dict(os.environ)
    env["DN_BENCH_NESTED"] = "1"
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-cpu-baseline", "--profile-steps", "0", "--alt-steps", "0",
           "--extras", "0"] + flags
    try:
        out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s, text=True)
    except subprocess.TimeoutExpired:
        return {"error": "timeout after %d s" % timeout_s}
    rows = [x for x in out.stdout.splitlines() if x.startswith("{")]
    if out.returncode != 0 or not rows:
        return {"error": "exit %d: %s" % (out.returncode, (out.stderr or "")[-300:])}
    return json.loads(rows[-1])


def run_extras(args):
    """What SURVEY.md section 8(d) / north_star ask beside the headline, timed by the SAME command the driver runs (round-5 verdict, item 3):
    `other_configs` = BASELINE configs[2], [4], [3] and the 480 x 640 step of the metric's model, 10 timed steps each;
    `strong_1gpu`   = the metric's step at 4 / 8 / 16 images (what one GPU of an 8 / 4 / 2-GPU data-parallel run of the literal metric,
    global batch 32, would compute) through the launch tape, and the same with the data-parallel machinery live on a single-rank own-RCCL
    communicator (gradient buckets, tape cuts, event fences, ncclAllReduce; only the wire is missing).  ONE-GPU figures, not a scaling
    curve.  Each entry: ms_per_step, img/s, the step's credited fraction of the 157.3 TFLOP/s fp32 matrix peak (BASELINE.md section 2
    FLOPs), or the error of that nested run."""
    def brief(l, extra=()):
        if "error" in l:
            return l
        gf = l.get("baseline_md_gflop_per_img")           # conv multiply-accumulates x 2 x 3 (forward + two gradients) per image
        r = {"img_per_s": l["value"], "ms_per_step": l["ms_per_step"], "ms_per_step_median": l.get("ms_per_step_median"),
             "steps": l["steps"],
             "step_credited_frac": (gf * l["value"] / 1e3 / PEAK_FP32_MFMA_TFLOPS) if gf else None,     # of the 157.3 TFLOP/s fp32 matrix peak
             "workload": l["config"]["workload"], "launch": l["config"]["launch"].split(":")[0], "final_loss": l["config"].get("final_loss")}
        for k in extra:
            r[k] = l["config"].get(k)
        return r
    other = {}
    for cfg in ("photo128", "dorn128", "res50_480", "vggbn480"):
        other[cfg] = brief(_nested(["--config", cfg, "--steps", "10", "--warmup", "3"]))
    strong = {"note": "one GPU computing 32 / N images per step (global batch 32 over N ranks): a bound, not a measured scaling curve"}
    for b in (4, 8, 16):
        strong["b%d_tape" % b] = brief(_nested(["--batch", str(b), "--steps", "40", "--warmup", "8", "--launch", "tape"]))
        strong["b%d_tape_rccl1" % b] = brief(_nested(["--batch", str(b), "--steps", "40", "--warmup", "8", "--launch", "tape", "--reducer", "rccl1"]),
                                             extra=("comm", "comm_buckets"))
    return other, strong


def start_watchdog(args, world, rank):
    """N > 1: a collective that never returns (a rank died, RCCL wedged) would leave the driver without any line.  After --watchdog-s
    seconds rank 0 prints ONE line that keeps the contract's keys (value null, the reason in `error`) and every rank leaves."""
    import threading
    limit = args.watchdog_s if args.watchdog_s > 0 else (900.0 if world > 1 else 0.0)
    if limit <= 0:
        return None

    def bail():
        if rank == 0:
            print(json.dumps({"metric": CONFIGS[args.config][0], "value": None, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
                              "dtype": args.compute, "data": "synthetic", "config": {"name": args.config, "parallelism": "dp%d" % world},
                              "error": "watchdog: no result after %.0f s (a rank or a collective is stuck); nothing was measured" % limit}))
            sys.stdout.flush()
        os._exit(3)

    t = threading.Timer(limit, bail)
    t.daemon = True
    t.start()
    return t


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-execute this command line as N ranks under torch.distributed.run
    (--nnodes=1, rendezvous on 127.0.0.1 and a port the kernel just handed out), stream the children's output through, and
    return their exit status.  LOCAL_RANK -> device is taken modulo the visible device count, so two ranks on a one-GPU box
    (DN_DIST_BACKEND=gloo) exercise the same path."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    if rc:
        raise SystemExit(rc)


def rccl_selfcheck(line, dev, rank, world, timeout_s=90.0):
    """N > 1 only, AFTER every timed region: one 20 MB gradient-bucket-sized buffer is summed over the ranks through this library's own
    RCCL communicator (rccl.Communicator: ncclCommInitRank from a unique id passed through torch.distributed's store, ncclAllReduce
    on a library stream, event fences) and through torch.distributed.all_reduce, and the two results are compared bit for bit; both
    paths are also timed (10 all-reduces each).  Recorded as config.rccl_selfcheck -- since round 6 the own communicator IS the default
    beyond one rank (behind distributed.open_rccl_communicator's preflight); this is the same comparison at bucket size, with timings.  A watchdog prints the line (rank 0) and ends the process if the own path blocks, so the headline never depends
    on it."""
    import threading
    import torch.distributed as dist
    result = {"status": "started", "world": world}
    if line is not None:
        line["config"]["rccl_selfcheck"] = result

    def bail():
        result["status"] = "timeout after %.0f s (own-RCCL path blocked); the line above it is unaffected" % timeout_s
        if rank == 0:
            print(json.dumps(line))
            sys.stdout.flush()
        os._exit(0)

    dog = threading.Timer(timeout_s, bail)
    dog.daemon = True
    dog.start()
    try:
        from supervised_dispnet_amd.distributed import open_rccl_communicator
        comm = open_rccl_communicator(dev)
        if comm is None:
            result["status"] = "own communicator not available on every rank"
            return
        n = (20 << 20) // 4
        g = torch.Generator().manual_seed(1234 + rank)
        src = torch.randn(n, generator=g).to(dev)
        a, b = src.clone(), src.clone()
        comm.all_reduce_sum_(a, [torch.cuda.current_stream()])
        comm.join()
        dist.all_reduce(b, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        same = bool(torch.equal(a, b))
        maxdiff = float((a - b).abs().max().item())
        times = {}
        for name in ("rccl-own", "torch.distributed"):
            x = src.clone()
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(10):
                if name == "rccl-own":
                    comm.all_reduce_sum_(x, [torch.cuda.current_stream()])
                    comm.join()
                else:
                    dist.all_reduce(x, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
            times[name] = (time.perf_counter() - t0) / 10 * 1e6
        ok = torch.tensor([1 if same else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        comm.destroy()
        result.update({"status": "ok", "bitwise_equal_on_all_ranks": bool(int(ok.item()) == 1), "max_abs_diff_rank0": maxdiff,
                       "bytes": n * 4, "us_per_allreduce": times})
    except Exception as e:                                     # noqa: BLE001 -- diagnostic only
        result["status"] = "error: %s: %s" % (type(e).__name__, str(e)[:200])
    finally:
        dog.cancel()


def dry_run(args, world, rank):
    """The N>1 plumbing without a GPU (CI, gloo): parameter arena in gradient-production order, bucketing, one all-reduce cycle
    driven through the same GradSink hooks the engine calls, the shard slice of the strong-scaling batch, tear-down."""
    import torch.distributed as dist
    import supervised_dispnet_amd.models as models
    from supervised_dispnet_amd import engine
    from supervised_dispnet_amd.distributed import GradReducer, shard_slice
    from supervised_dispnet_amd.optim import ParamArena
    if world > 1:
        dist.init_process_group("gloo")
    torch.manual_seed(0)
    net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
    _quiet_init(net)
    arena = ParamArena(list(net._hot_parameters()), net._grad_production_order())
    reducer = GradReducer(arena)
    engine.GradSink.reducer = reducer
    sink = engine.GradSink()
    for p in arena.params:                                   # what the engine does as gradients land, decoder first
        p._dn_grad_view.fill_(float(rank + 1))
        sink.put(p, p._dn_grad_view)
    scale = reducer.finish()
    want = sum(range(1, world + 1))
    ok = abs(scale - 1.0 / world) < 1e-12
    for p in arena.params:
        ok = ok and bool(torch.all(p._dn_grad_view == want))
    sl = shard_slice(args.global_batch, rank, world)
    engine.GradSink.reducer = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "ok": ok, "world": world, "buckets": len(reducer.buckets),
                          "bucket_mb": [round((b["hi"] - b["lo"]) * 4 / 2 ** 20, 1) for b in reducer.buckets],
                          "arena_params": len(arena.params), "shard": [sl.start, sl.stop]}))
    if not ok:
        raise SystemExit(1)


PMC_CONFIG = "vggbn128"     # set by main(): which committed PMC summary describes this run's workload


def _pmc_file():
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if PMC_CONFIG == "vggbn128":        # rNN_x_pmc_traffic.json; the other configs' summaries carry the config name
        files = [f for f in files if re.match(r"r\d+_[a-z]+_pmc_traffic\.json$", os.path.basename(f))]
    else:
        files = [f for f in files if os.path.basename(f).endswith("_%s_pmc_traffic.json" % PMC_CONFIG)]
    return files[-1] if files else None


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` (read + write) from the committed rocprofv3 PMC passes (tools/pmc_traffic.sh: separate
    FETCH_SIZE / WRITE_SIZE runs of this same command, corrected and calibrated as profiles/*pmc_traffic.json states), or None.
    The counters cannot be collected from inside this process, so the latest committed summary of this config is reported."""
    f = _pmc_file()
    if f is None:
        return None
    try:
        k = json.load(open(f))["kernels"].get(kernel)
        return (k["read_bytes"] + k["write_bytes"]) if k else None
    except Exception:
        return None


def pmc_traffic_source(kernel):
    """Which committed profile `traffic` was read from: the PMC passes describe the build that was PROFILED, which may be older than
    the build this run timed."""
    f = _pmc_file()
    if f is None or pmc_traffic(kernel) is None:
        return None
    return "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the build that was profiled, not measured in this run)" % os.path.basename(f)


def _quiet_init(net):
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        net.init_weights(use_pretrained_weights=False)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-replica":
        _cpu_replica_main(sys.argv[2])
    else:
        main()
