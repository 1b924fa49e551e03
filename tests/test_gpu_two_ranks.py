"""Two data-parallel ranks, really running, on the ONE GPU this pool offers: two processes on cuda:0, torch.distributed over gloo on
DEVICE tensors (RCCL refuses two ranks on one device), DN_COMM=torch.  pytest -m gpu.

What it pins (reference train.py:316-317 = nn.DataParallel over b32):
  * three training steps of Disp_vgg_BN at 2 ranks x b2 equal ONE process stepping the same four images with per-rank BatchNorm
    statistics (two replicas sharing weights, each forwarding its shard; gradients added; one Adam step) -- bit for bit with
    l1_loss (per-sample means: nothing but the gradient sum crosses ranks, and a + b == b + a), and to fp32 summation order with
    Multiscale_L1 (train.py's DEFAULT --loss), whose whole-batch (sum, count) statistics are exchanged in the FORWARD pass while the
    gradient buckets travel in the BACKWARD pass: both exchanges interleave on every step, on both ranks;
  * both ranks end with bit-identical parameters.
"""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W, PER_RANK, STEPS = 64, 96, 2, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _batch():
    g = torch.Generator().manual_seed(3)
    n = 2 * PER_RANK
    img = (torch.rand(n, 3, H, W, generator=g) - 0.5) / 0.5
    depth = torch.rand(n, H, W, generator=g) * 79.0 + 1.0
    keep = torch.rand(n, H, W, generator=g) < torch.tensor([0.05, 0.2, 0.5, 0.1]).view(n, 1, 1)     # very different valid counts per rank
    return img, depth * keep.float()


def _make_net(dev):
    import bench
    import supervised_dispnet_amd.models as models
    from supervised_dispnet_amd.optim import FusedAdam
    torch.manual_seed(0)
    net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
    bench._quiet_init(net)
    net.to(dev).train()
    opt = FusedAdam(net._hot_parameters(), lr=1e-4, production_order=net._grad_production_order())
    return net, opt


def _loss(LF, name, gt, depth):
    return LF.l1_loss(gt, depth, "kitti") if name == "L1" else LF.Multiscale_L1_loss(gt, depth)


def _worker(rank, world, port, loss_name, out_dir, taped=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DN_COMM="torch", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        import supervised_dispnet_amd.loss_functions as LF
        from supervised_dispnet_amd import distributed as DD, engine
        from supervised_dispnet_amd.functional import reciprocal
        net, opt = _make_net(dev)
        red = DD.GradReducer(opt.arena, bucket_bytes=4 << 20)
        assert red.path == "torch.distributed:gloo" and red.world == 2 and len(red.buckets) >= 4
        engine.GradSink.reducer = red
        DD.ISSUE_LOG = []
        img, gt = _batch()
        sl = DD.shard_slice(img.shape[0], rank, world)
        img, gt = img[sl].to(dev), gt[sl].to(dev)
        losses, g1 = [], None
        if taped:
            # the same steps through the launch tape: step 1 is recorded (a real step), steps 2.. are replays -- the tape is cut at
            # every gradient bucket and the all-reduce runs live between the segments (engine.tape_host_call)
            from supervised_dispnet_amd.graph import TapedStep, backward
            scale = 0.5

            def step():
                depth = [reciprocal(d) for d in net(img)]
                loss = _loss(LF, loss_name, gt, depth)
                opt.zero_grad()
                backward(loss)
                opt.step(grad_scale=red.finish())
                return loss

            opt.overlap_backward(red)          # + the Adam update of each bucket right behind its all-reduce (FusedAdam.overlap_backward)
            ts = TapedStep(step, optimizer=opt, warmup=0)
            for it in range(STEPS):
                losses.append(float(ts().item()))
            assert ts.segments == len(red.buckets) + 2 and len(ts.host_calls) == len(red.buckets) + 1, (ts.segments, len(red.buckets))
            g1 = opt.arena.flat_g.clone()            # (after the last step: only compared between the ranks)
        for it in range(0 if taped else STEPS):
            depth = [reciprocal(d) for d in net(img)]
            loss = _loss(LF, loss_name, gt, depth)
            opt.zero_grad()
            loss.backward()
            scale = red.finish()
            if it == 0:
                torch.cuda.synchronize()
                g1 = opt.arena.flat_g.clone()
            opt.step(grad_scale=scale)
            losses.append(float(loss.item()))
        torch.cuda.synchronize()
        torch.save({"p": opt.arena.flat_p.cpu(), "g1": g1.cpu(), "losses": losses, "log": DD.ISSUE_LOG, "scale": scale},
                   os.path.join(out_dir, "rank%d.pt" % rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _single_process_reference(loss_name):
    """The same four images in ONE process: two replicas with identical weights (per-rank BatchNorm statistics), gradients added."""
    import supervised_dispnet_amd.loss_functions as LF
    from supervised_dispnet_amd.functional import reciprocal
    dev = torch.device("cuda:0")
    reps = [_make_net(dev) for _ in range(2)]
    assert torch.equal(reps[0][1].arena.flat_p, reps[1][1].arena.flat_p)
    img, gt = _batch()
    img, gt = img.to(dev), gt.to(dev)
    losses, g1 = [], None
    for it in range(STEPS):
        if loss_name == "L1":
            # per-sample means / B: every rank's loss is its own shard's; the data-parallel step averages the ranks' gradients
            tot = None
            for r, (net, opt) in enumerate(reps):
                sl = slice(r * PER_RANK, (r + 1) * PER_RANK)
                depth = [reciprocal(d) for d in net(img[sl])]
                loss = _loss(LF, loss_name, gt[sl], depth)
                opt.zero_grad()
                loss.backward()
                tot = opt.arena.flat_g.clone() if tot is None else tot + opt.arena.flat_g
            scale = 0.5
            losses.append(float(loss.item()))                                  # rank 1's (the last shard's) loss
        else:
            # whole-batch loss: one value over all four images (what the reference's DataParallel computes on the gathered batch)
            outs = [[reciprocal(d) for d in net(img[r * PER_RANK:(r + 1) * PER_RANK])] for r, (net, _o) in enumerate(reps)]
            depth = [torch.cat([outs[0][s], outs[1][s]], 0) for s in range(4)]
            loss = _loss(LF, loss_name, gt, depth)
            for _n, opt in reps:
                opt.zero_grad()
            loss.backward()
            tot = reps[0][1].arena.flat_g + reps[1][1].arena.flat_g
            scale = 1.0
            losses.append(float(loss.item()))
        if it == 0:
            g1 = (tot * scale * 2.0).clone()                                    # what the ranks' summed arena holds (x world for Multi_L1)
        for _n, opt in reps:
            opt.arena.flat_g.copy_(tot)
            opt.step(grad_scale=scale)
    torch.cuda.synchronize()
    return reps[0][1].arena.flat_p.cpu(), g1.cpu(), losses


@pytest.mark.parametrize("loss_name", ["L1", "Multi_L1"])
def test_two_ranks_on_one_gpu_equal_the_single_process_step(loss_name, tmp_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, loss_name, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    r0 = torch.load(os.path.join(str(tmp_path), "rank0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "rank1.pt"))
    # both ranks: same summed gradients, same parameters, same collective issue order
    assert torch.equal(r0["g1"], r1["g1"]) and torch.equal(r0["p"], r1["p"])
    assert r0["log"] == r1["log"] and r0["scale"] == 0.5
    nb = len([e for e in r0["log"] if e[0] == "bucket"]) // STEPS
    ns = len([e for e in r0["log"] if e[0] == "stats"]) // STEPS
    assert nb >= 4 and ns == (0 if loss_name == "L1" else 2)                    # Multi_L1: one sum + one max exchange per forward
    if loss_name == "Multi_L1":
        assert r0["losses"] == r1["losses"]                                     # whole-batch value: identical on every rank
        first = [e[0] for e in r0["log"][:ns + nb]]
        assert first[:ns] == ["stats"] * ns and set(first[ns:]) == {"bucket"}   # forward exchange, then the backward's buckets
    p_ref, g_ref, l_ref = _single_process_reference(loss_name)
    if loss_name == "L1":
        assert r1["losses"] == l_ref
        assert torch.equal(r0["g1"], g_ref), float((r0["g1"] - g_ref).abs().max())
        assert torch.equal(r0["p"], p_ref), float((r0["p"] - p_ref).abs().max())
    else:
        # (sum, count) added across ranks vs reduced over the whole batch by one kernel: fp32 summation order
        assert r0["losses"][0] == pytest.approx(l_ref[0], rel=2e-6)
        gs = float(g_ref.abs().max())
        assert float((r0["g1"] - g_ref).abs().max()) <= 2e-5 * gs, (float((r0["g1"] - g_ref).abs().max()), gs)
        # Adam turns a gradient into ~lr * sign(g): an element whose gradient is within the 1e-5-level difference of zero may move the
        # other way (|dp| <= 2 * lr per step); everything else follows to ~1e-9
        dp = (r0["p"] - p_ref).abs()
        assert float(dp.max()) <= 2.1e-4 * STEPS, float(dp.max())
        assert float((dp > 2e-6).float().mean()) <= 2e-3, float((dp > 2e-6).float().mean())


def test_two_ranks_through_the_launch_tape_equal_the_eager_two_rank_run(tmp_path):
    """graph.TapedStep under data parallelism: the recorded step is cut at every gradient bucket (engine.tape_host_call) and the
    all-reduce is issued live between the replayed segments, each followed by the Adam update of that bucket on the optimizer's own
    stream (FusedAdam.overlap_backward).  Two ranks on one GPU, l1_loss, three steps (one recorded + two replays):
    same collective issue order as the eager run on both ranks, bit-identical parameters across the ranks, and the eager two-rank
    result up to the last bit of Adam's step size (device-side vs host-side pow() of the bias corrections, as in the hipGraph test)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    res = {}
    for taped in (False, True):
        out = tmp_path / ("taped" if taped else "eager")
        out.mkdir()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, "L1", str(out), taped)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=600)
        assert all(p.exitcode == 0 for p in procs), (taped, [p.exitcode for p in procs])
        res[taped] = [torch.load(os.path.join(str(out), "rank%d.pt" % r)) for r in range(2)]
    t0, t1 = res[True]
    e0 = res[False][0]
    assert torch.equal(t0["p"], t1["p"]) and torch.equal(t0["g1"], t1["g1"])
    assert t0["log"] == t1["log"] == e0["log"]                                   # same buckets, same order, every step
    assert t0["losses"][0] == e0["losses"][0]
    for lt, le in zip(t0["losses"], e0["losses"]):
        assert lt == pytest.approx(le, rel=1e-5)
    assert torch.allclose(t0["p"], e0["p"], rtol=0, atol=2e-7), float((t0["p"] - e0["p"]).abs().max())


def test_bench_plain_gpus_2_spawns_both_ranks_and_prints_one_line():
    """`python bench.py --gpus 2 ...` with no launcher (what a driver's plain scale run would be): bench.py becomes its own
    torch.distributed.run launcher; both ranks land on the one GPU (LOCAL_RANK modulo the device count), gloo carries the gradient
    buckets, rank 0 prints the ONE JSON line with n_gpus 2, the strong-scaling shard (32 / 2 per rank) and config.comm."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DN_DIST_BACKEND="gloo", DN_COMM="torch")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--alt-steps", "0",
                        "--profile-steps", "0", "--global-batch", "8"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_batch"] == 8
    assert d["config"]["parallelism"] == "dp2" and d["config"]["comm"] not in (None, "none (one rank)")
    assert d["value"] > 0 and d["steps"] == 3
