"""The whole training step replayed as one captured hipGraph (supervised_dispnet_amd/graph.py) is the SAME arithmetic as the
eager launch sequence: after k steps from identical initial state the parameters, the Adam moments, the BatchNorm running
statistics and the per-step losses agree (bit for bit up to the last bit of Adam's step size, see below), with new batches written into the static input buffers between replays."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import bench  # noqa: E402
import supervised_dispnet_amd.loss_functions as LF  # noqa: E402
import supervised_dispnet_amd.models as models  # noqa: E402
from supervised_dispnet_amd.functional import reciprocal  # noqa: E402
from supervised_dispnet_amd.graph import GraphedStep  # noqa: E402
from supervised_dispnet_amd.optim import FusedAdam  # noqa: E402

DEV = torch.device("cuda:0")


def _make(sd0=None):
    torch.manual_seed(0)
    net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
    bench._quiet_init(net)
    if sd0 is not None:
        net.load_state_dict(sd0)
    net.to(DEV).train()
    opt = FusedAdam(net._hot_parameters(), lr=1e-4, production_order=net._grad_production_order())
    return net, opt


@pytest.mark.parametrize("batch,h,w", [(4, 128, 416), (2, 64, 96)], ids=["b4_128x416", "b2_64x96"])
def test_graph_replay_is_bitwise_the_eager_step(batch, h, w):
    batches = [bench.synthetic_batch(batch, h, w, DEV, seed) for seed in range(5)]
    net_e, opt_e = _make()
    sd0 = copy.deepcopy({k: v.detach().cpu().clone() for k, v in net_e.state_dict().items()})
    net_g, opt_g = _make(sd0)

    img, gt = batches[0][0].clone(), batches[0][1].clone()          # the graph's static input buffers

    def step_g():
        depth = [reciprocal(d) for d in net_g(img)]
        loss = LF.l1_loss(gt, depth, "kitti")
        opt_g.zero_grad()
        loss.backward()
        opt_g.step()
        return loss

    def step_e(x, y):
        depth = [reciprocal(d) for d in net_e(x)]
        loss = LF.l1_loss(y, depth, "kitti")
        opt_e.zero_grad()
        loss.backward()
        opt_e.step()
        return loss

    gs = GraphedStep(step_g, optimizer=opt_g, warmup=2, static_inputs=(img, gt))
    # the two warm-up iterations inside capture() are real steps on batch 0: mirror them (and the captured one never runs)
    gs.capture()
    for _ in range(2):
        step_e(*batches[0])
    losses_g, losses_e = [], []
    for x, y in batches:
        img.copy_(x)
        gt.copy_(y)
        losses_g.append(gs().clone())
        losses_e.append(step_e(x, y).clone())
    torch.cuda.synchronize()
    assert int(opt_g._dev["step"].item()) == opt_e.step_count == 7
    assert torch.equal(losses_g[0], losses_e[0])
    for lg, le in zip(losses_g, losses_e):
        assert torch.allclose(lg, le, rtol=1e-5)
    # the bias corrections are evaluated by the device's pow() in the graph and by the host's in the eager form: step_size may
    # differ in its last bit, i.e. by 1e-11 per update of 1e-4 -- everything else is the same instruction stream
    assert torch.allclose(opt_g.arena.flat_p, opt_e.arena.flat_p, rtol=0, atol=2e-7)
    assert torch.allclose(opt_g.exp_avg, opt_e.exp_avg, rtol=1e-4, atol=1e-9)
    sg, se = net_g.state_dict(), net_e.state_dict()
    for k in sg:
        assert torch.allclose(sg[k].float(), se[k].float(), rtol=1e-5, atol=2e-7), k
    # the optimizer state round-trips with the device-side counter
    assert opt_g.state_dict()["step"] == 7


def test_bn_backward_without_materialised_dz_is_bitwise_the_two_pass_form(monkeypatch):
    """engine.BN_MATERIALIZE_DZ: the reduce pass that only produces the sums + the apply pass that re-derives the ReLU mask / expands
    the max-pool routing (dn_bn_relu_bwd_sums, dn_bn_relu_pool_bwd_sums, dn_bn_bwd_apply_relu, dn_bn_bwd_apply_pool) is the same
    arithmetic as the r01 form that writes dz and reads it back: every gradient of a Disp_vgg_BN step agrees bit for bit."""
    from supervised_dispnet_amd import engine
    img, gt = bench.synthetic_batch(4, 128, 416, DEV, 3)
    grads = {}
    # (the sums taken in the input-gradient epilogues, round 3, are the same numbers in another summation order: that fusion has its own
    #  test, tests/test_gpu_kernels.py::test_bn_backward_sums_fused_into_the_input_gradient, and is switched off for this bitwise one)
    monkeypatch.setattr(engine, "BN_SUMS_FUSION", False)
    for mode in (False, True):
        monkeypatch.setattr(engine, "BN_MATERIALIZE_DZ", mode)
        net, opt = _make()
        depth = [reciprocal(d) for d in net(img)]
        loss = LF.l1_loss(gt, depth, "kitti")
        opt.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        grads[mode] = (loss.item(), opt.arena.flat_g.clone())
    assert grads[False][0] == grads[True][0]
    assert torch.equal(grads[False][1], grads[True][1])


@pytest.mark.parametrize("batch,h,w", [(4, 128, 416), (2, 64, 96)], ids=["b4_128x416", "b2_64x96"])
def test_launch_tape_replay_is_bitwise_the_eager_step(batch, h, w):
    """graph.TapedStep (dn_tape_*): the launches and stream fences of one recorded step, re-issued by one C call per step, are the same
    arithmetic as the eager launch sequence -- losses, parameters, Adam moments and BatchNorm buffers after 7 steps on changing batches
    (written into the static input buffers between replays).  A launch the tape missed (framework-side work) or a pointer that moved
    would show up here as stale data."""
    from supervised_dispnet_amd.graph import TapedStep, backward
    batches = [bench.synthetic_batch(batch, h, w, DEV, seed) for seed in range(5)]
    net_e, opt_e = _make()
    sd0 = copy.deepcopy({k: v.detach().cpu().clone() for k, v in net_e.state_dict().items()})
    net_t, opt_t = _make(sd0)
    img, gt = batches[0][0].clone(), batches[0][1].clone()

    def step_t():
        depth = [reciprocal(d) for d in net_t(img)]
        loss = LF.l1_loss(gt, depth, "kitti")
        opt_t.zero_grad()
        backward(loss)
        opt_t.step()
        return loss

    def step_e(x, y):
        depth = [reciprocal(d) for d in net_e(x)]
        loss = LF.l1_loss(y, depth, "kitti")
        opt_e.zero_grad()
        loss.backward()
        opt_e.step()
        return loss

    ts = TapedStep(step_t, optimizer=opt_t, warmup=2, static_inputs=(img, gt))
    ts.capture()                       # 2 warm-up steps + the recorded one: three real steps on batch 0
    for _ in range(3):
        step_e(*batches[0])
    assert ts.launches > 100 and ts.fences > 10 and ts.segments == 1
    losses_t, losses_e = [], []
    for x, y in batches[1:]:
        img.copy_(x)
        gt.copy_(y)
        losses_t.append(ts().clone())
        losses_e.append(step_e(x, y).clone())
    # scribble over freed memory of the regular pool between replays: the tape's buffers live in their own pool
    junk = [torch.full((1 << 20,), float("nan"), device=DEV) for _ in range(8)]
    del junk
    img.copy_(batches[0][0])
    gt.copy_(batches[0][1])
    losses_t.append(ts().clone())
    losses_e.append(step_e(*batches[0]).clone())
    torch.cuda.synchronize()
    assert int(opt_t._dev["step"].item()) == opt_e.step_count == 8
    for lt, le in zip(losses_t, losses_e):
        assert torch.isfinite(lt).all()
        assert torch.allclose(lt, le, rtol=1e-5), (lt, le)
    assert torch.equal(losses_t[0], losses_e[0])
    # (device pow() vs host pow() in Adam's bias corrections: see the hipGraph test above)
    assert torch.allclose(opt_t.arena.flat_p, opt_e.arena.flat_p, rtol=0, atol=2e-7)
    assert torch.allclose(opt_t.exp_avg, opt_e.exp_avg, rtol=1e-4, atol=1e-9)
    st, se = net_t.state_dict(), net_e.state_dict()
    for k in st:
        assert torch.allclose(st[k].float(), se[k].float(), rtol=1e-5, atol=2e-7), k


@pytest.mark.parametrize("taped", [False, True], ids=["eager", "taped"])
def test_adam_per_bucket_under_the_backward_pass_is_bitwise_the_single_update(taped):
    """FusedAdam.overlap_backward: the update of a gradient bucket is enqueued on the optimizer's own stream as soon as the bucket's
    gradients are complete (distributed.GradReducer watches the buckets; one rank: nothing is exchanged) and runs under the rest of
    the backward pass.  Element-wise arithmetic: parameters, moments and losses equal the single whole-arena update bit for bit --
    eagerly, and with the whole step (ranges, tick, fences against both compute streams) on a launch tape."""
    from supervised_dispnet_amd import engine
    from supervised_dispnet_amd.distributed import GradReducer
    from supervised_dispnet_amd.graph import TapedStep, backward
    batches = [bench.synthetic_batch(4, 64, 96, DEV, seed) for seed in range(4)]
    net_a, opt_a = _make()
    sd0 = copy.deepcopy({k: v.detach().cpu().clone() for k, v in net_a.state_dict().items()})
    net_b, opt_b = _make(sd0)
    opt_a.capturable(True)
    opt_b.capturable(True)
    red = GradReducer(opt_b.arena, bucket_bytes=8 << 20, comm="torch")
    assert red.path == "none" and len(red.buckets) >= 4
    opt_b.overlap_backward(red)
    img, gt = batches[0][0].clone(), batches[0][1].clone()

    def step_a(x, y):
        depth = [reciprocal(d) for d in net_a(x)]
        loss = LF.l1_loss(y, depth, "kitti")
        opt_a.zero_grad()
        backward(loss)
        opt_a.step()
        return loss

    def step_b():
        engine.GradSink.reducer = red
        try:
            depth = [reciprocal(d) for d in net_b(img)]
            loss = LF.l1_loss(gt, depth, "kitti")
            opt_b.zero_grad()
            backward(loss)
            opt_b.step(grad_scale=red.finish())
        finally:
            engine.GradSink.reducer = None
        return loss

    f = TapedStep(step_b, optimizer=opt_b, warmup=1, static_inputs=(img, gt)) if taped else step_b
    if taped:
        f.capture()                    # one warm-up + the recorded step on batch 0
        for _ in range(2):
            step_a(*batches[0])
        assert f.segments == 1
    for x, y in batches:
        img.copy_(x)
        gt.copy_(y)
        lb = f().clone()
        la = step_a(x, y).clone()
        assert torch.equal(la, lb)
    torch.cuda.synchronize()
    assert int(opt_a._dev["step"].item()) == int(opt_b._dev["step"].item()) == (6 if taped else 4)
    assert torch.equal(opt_a.arena.flat_p, opt_b.arena.flat_p)
    assert torch.equal(opt_a.exp_avg, opt_b.exp_avg) and torch.equal(opt_a.exp_avg_sq, opt_b.exp_avg_sq)


@pytest.mark.parametrize("riding", [True, False], ids=["riding-fences", "own-event-records"])
def test_launch_tape_lazy_join_runs_replays_back_to_back_and_joins_on_demand(monkeypatch, riding):
    """TapedStep(lazy_join=True): a replay does not make the caller's stream wait for the tape's (bench.py: ~25 us of idle device between
    two steps saved); join() does.  Four steps on changing batches written into the static inputs on the CALLER's stream: losses (read
    after join()), parameters and Adam moments equal the eager steps' bit for bit -- the input fence in front of every replay still orders
    the caller's copies before the step, and the fences between the compute streams are device-scope events (dn_tape_fence_device) --
    re-issued as the stop event of the launch in front of them (hipExtLaunchKernel; `riding`), or, with DN_NO_RIDING_FENCES=1, as event
    records of their own."""
    from supervised_dispnet_amd import _lib
    from supervised_dispnet_amd.graph import TapedStep, backward
    if not riding:
        monkeypatch.setenv("DN_NO_RIDING_FENCES", "1")
        _lib.load().dn_reload_knobs()
    batches = [bench.synthetic_batch(4, 64, 96, DEV, seed) for seed in range(4)]
    net_a, opt_a = _make()
    sd0 = copy.deepcopy({k: v.detach().cpu().clone() for k, v in net_a.state_dict().items()})
    net_b, opt_b = _make(sd0)
    opt_a.capturable(True)
    img, gt = batches[0][0].clone(), batches[0][1].clone()

    def step_a(x, y):
        depth = [reciprocal(d) for d in net_a(x)]
        loss = LF.l1_loss(y, depth, "kitti")
        opt_a.zero_grad()
        backward(loss)
        opt_a.step()
        return loss

    def step_b():
        depth = [reciprocal(d) for d in net_b(img)]
        loss = LF.l1_loss(gt, depth, "kitti")
        opt_b.zero_grad()
        backward(loss)
        opt_b.step()
        return loss

    f = TapedStep(step_b, optimizer=opt_b, warmup=0, static_inputs=(img, gt), lazy_join=True).capture()     # the recorded step: batch 0
    assert f.lazy_join and f.fences > 0
    assert (f.riding_fences > 0) == riding and f.riding_fences <= f.fences
    la = [step_a(*batches[0]).clone()]
    lb = []
    for x, y in batches[1:]:
        img.copy_(x)                    # on the caller's stream; the replay's input fence orders it
        gt.copy_(y)
        out = f()
        f.join()
        lb.append(out.clone())
        la.append(step_a(x, y).clone())
    # back to back, one join at the end: the inputs are refilled ON THE TAPE'S STREAM (write_inputs) -- a copy_ on the caller's stream
    # here would race with the previous replay's reads of img / gt (its weight gradient and loss read them last), nothing orders the two
    for x, y in batches[:2]:
        f.write_inputs(x, y)
        out = f()
        step_a(x, y)
    f.join()
    with pytest.raises(ValueError):
        f.write_inputs(x)
    torch.cuda.synchronize()
    for u, v in zip(la[1:], lb):
        assert torch.equal(u, v)
    assert torch.equal(opt_a.arena.flat_p, opt_b.arena.flat_p)
    assert torch.equal(opt_a.exp_avg, opt_b.exp_avg) and torch.equal(opt_a.exp_avg_sq, opt_b.exp_avg_sq)


def test_riding_fence_behind_a_kernel_with_140_kb_of_dynamic_lds():
    """A device-scope tape fence whose waitee's last item is a launch is re-issued as that launch's STOP event (hipExtLaunchKernel).  Here the
    launch is the 8-wave Winograd kernel (140 KB of dynamic LDS: the attribute the plain launches set must hold for the extended launch
    too) and the waiting stream copies its output: the replay on NEW input equals a fresh eager convolution bit for bit, twice."""
    from supervised_dispnet_amd import _lib, engine
    from supervised_dispnet_amd.graph import TapedStep
    import torch.nn as nn
    torch.manual_seed(5)
    N, H, W, Cc = 4, 64, 208, 128
    layer = engine.ConvLayer(nn.Conv2d(Cc, Cc, 3, 1, 1).to(DEV))
    x = torch.randn(N, H, W, Cc, device=DEV)
    act = engine.Act(x, N, H, W, Cc)
    out = torch.zeros(N, H, W, Cc, device=DEV)
    names = []

    def step():
        with engine.stream_scope():
            y, _, _ = engine.conv_forward(layer, [engine.Piece(act)])
            names.append(_lib.load().dn_last_kernel().decode())
            main, side = torch.cuda.current_stream(), engine.side_stream()["sides"][0]
            engine.stream_wait(side, main, device_scope=True)
            _lib.call("dn_copy", y.data_ptr(), out.data_ptr(), y.numel(), side.cuda_stream)
            engine.cross_stream_use(y, side)
            engine.stream_wait(main, side, device_scope=True)
        return out

    ts = TapedStep(step, warmup=1).capture()
    assert "wino_conv8_kernel" in names[-1], names[-1]
    assert ts.fences == 2 and ts.riding_fences == 2
    for seed in (1, 2):
        x.copy_(torch.randn(N, H, W, Cc, device=DEV, generator=torch.Generator(device=DEV).manual_seed(seed)))
        out.zero_()
        got = ts().clone()
        torch.cuda.synchronize()
        want, _, _ = engine.conv_forward(layer, [engine.Piece(act)])
        torch.cuda.synchronize()
        assert torch.equal(got, want)


def test_launch_tape_refuses_framework_side_device_work_and_recovers():
    """A step that does device work outside libdispnet_hip while a tape is recorded (here: a gradient seeded into an activation that
    already holds one -- an ATen add) fails the recording loudly instead of producing a tape that would silently skip it; the recording
    slot is free again afterwards and eager steps still run."""
    from supervised_dispnet_amd import engine
    from supervised_dispnet_amd.graph import TapedStep, backward
    img, gt = bench.synthetic_batch(2, 64, 96, DEV, 0)
    net, opt = _make()

    def good():
        depth = [reciprocal(d) for d in net(img)]
        loss = LF.l1_loss(gt, depth, "kitti")
        opt.zero_grad()
        backward(loss)
        opt.step()
        return loss

    def bad():
        loss = good()
        engine._not_on_tape("a framework-side kernel")
        return loss

    with pytest.raises(RuntimeError, match="launch tape"):
        TapedStep(bad, optimizer=opt, warmup=0).capture()
    assert engine.TAPE is None
    l0 = good()                                         # eager still works
    ts = TapedStep(good, optimizer=opt, warmup=0).capture()      # and the slot is free
    l1 = ts()
    torch.cuda.synchronize()
    assert torch.isfinite(l0) and torch.isfinite(l1) and float(l1) < float(l0)


def test_launch_tape_outlives_other_models_in_the_weight_relay_table():
    """The batched weight re-lay (engine.PackTable) is per device, not per model: a step recorded while ANOTHER model is alive re-lays
    that model's rows too.  The tape keeps what it recorded alive (device tables, weights, packed buffers), so deleting the other model
    -- and letting the allocator hand its memory out again -- must neither fault nor change the taped model's results."""
    import gc
    from supervised_dispnet_amd.graph import TapedStep, backward
    batches = [bench.synthetic_batch(2, 64, 96, DEV, seed) for seed in range(4)]
    net_o, opt_o = _make()                                   # the "other" model: two eager steps register its rows
    for _ in range(2):
        depth = [reciprocal(d) for d in net_o(batches[0][0])]
        loss = LF.l1_loss(batches[0][1], depth, "kitti")
        opt_o.zero_grad()
        loss.backward()
        opt_o.step()
    net_e, opt_e = _make()
    sd0 = copy.deepcopy({k: v.detach().cpu().clone() for k, v in net_e.state_dict().items()})
    net_t, opt_t = _make(sd0)
    opt_e.capturable(True)
    img, gt = batches[0][0].clone(), batches[0][1].clone()

    def step_t():
        depth = [reciprocal(d) for d in net_t(img)]
        loss = LF.l1_loss(gt, depth, "kitti")
        opt_t.zero_grad()
        backward(loss)
        opt_t.step()
        return loss

    def step_e(x, y):
        depth = [reciprocal(d) for d in net_e(x)]
        loss = LF.l1_loss(y, depth, "kitti")
        opt_e.zero_grad()
        backward(loss)
        opt_e.step()
        return loss

    ts = TapedStep(step_t, optimizer=opt_t, warmup=1, static_inputs=(img, gt)).capture()     # net_o's rows are in the recorded re-lay
    for _ in range(2):
        step_e(*batches[0])
    del net_o, opt_o, depth, loss
    gc.collect()
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 22,), float("nan"), device=DEV) for _ in range(16)]             # reuse whatever was released
    for x, y in batches[1:]:
        img.copy_(x)
        gt.copy_(y)
        lt = ts().clone()
        le = step_e(x, y).clone()
        torch.cuda.synchronize()
        assert torch.isfinite(lt) and torch.equal(lt, le)
    del junk
    assert torch.equal(opt_t.arena.flat_p, opt_e.arena.flat_p)
    ts.close()
    ts.close()                                                # idempotent


def test_folded_reductions_are_bitwise_the_separate_kernels(monkeypatch):
    """engine.FOLD_FINALIZE (round 5, csrc/dn_fold.h): with few partial rows the Winograd forward finishes the BatchNorm statistics and the
    Winograd input gradient finishes the BatchNorm-backward sums in the block that arrives last, instead of dn_bn_finalize / the sums
    launch of dn_bn_bwd_apply_relu.  The stand-alone kernels use the same sliced order for up to 128 partial rows, so loss, outputs,
    EVERY gradient, the BatchNorm buffers and the parameters after two Adam steps are bit-identical with the folds on and off -- at 4
    images (the 8-GPU shard of the metric: nine forward + six backward folds) and at 2 x 64 x 96; and the folds really happen
    (fewer dn_bn_finalize calls, sums launches skipped)."""
    from oracle import detgen
    from supervised_dispnet_amd import _lib, engine
    for (b, h, w, fwd_folds, bwd_folds) in ((4, 128, 416, 9, 6), (2, 64, 96, 1, 1)):
        x = detgen.image_batch(b, h, w, "fold:x").to(DEV)
        gt = detgen.sparse_depth(b, h, w, "fold:gt", density=0.3).to(DEV)
        res, calls = {}, {}
        real_call = _lib.call
        for fold in (True, False):
            counts = {"dn_bn_finalize": 0, "sums_skipped": 0}

            def counting(name, *args, _c=counts):
                if name == "dn_bn_finalize":
                    _c["dn_bn_finalize"] += 1
                if name == "dn_bn_bwd_apply_relu" and args[7] is None:
                    _c["sums_skipped"] += 1
                return real_call(name, *args)

            monkeypatch.setattr(engine, "FOLD_FINALIZE", fold)
            monkeypatch.setattr(_lib, "call", counting)
            net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
            detgen.fill_state_dict(net.state_dict(), "vggbn")
            net.to(DEV).train()
            opt = FusedAdam(net._hot_parameters(), lr=1e-4, betas=(0.9, 0.999), production_order=net._grad_production_order())
            losses = []
            for _ in range(2):
                disps = net(x)
                loss = LF.l1_loss(gt, [reciprocal(d) for d in disps], "kitti")
                opt.zero_grad()
                loss.backward()
                grads = opt.arena.flat_g.clone()
                opt.step()
                losses.append(loss.item())
            torch.cuda.synchronize()
            monkeypatch.setattr(_lib, "call", real_call)
            res[fold] = (losses, [d.detach().clone() for d in disps], grads, opt.arena.flat_p.clone(),
                         {k: v.clone() for k, v in net.state_dict().items() if "running" in k or "num_batches" in k})
            calls[fold] = counts
        assert calls[False]["dn_bn_finalize"] == 2 * 13 and calls[False]["sums_skipped"] == 0
        assert calls[True]["dn_bn_finalize"] <= 2 * (13 - fwd_folds), calls
        assert calls[True]["sums_skipped"] >= 2 * bwd_folds, calls
        assert res[True][0] == res[False][0]
        assert all(torch.equal(a, c) for a, c in zip(res[True][1], res[False][1]))
        assert torch.equal(res[True][2], res[False][2]) and torch.equal(res[True][3], res[False][3])
        assert res[True][4].keys() == res[False][4].keys() and all(torch.equal(res[True][4][k], res[False][4][k]) for k in res[True][4])
        assert int(res[True][4]["features.features.1.num_batches_tracked"]) == 2
