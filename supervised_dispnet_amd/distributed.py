"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed backend "nccl" on ROCm) over xGMI.

The reference uses single-process nn.DataParallel (train.py:316-317): broadcast all parameters, gather outputs and
reduce gradients to GPU0 every iteration.  Here every rank owns a replica and the only exchange is ONE sum all-reduce of
the 79.5 MB gradient arena per step, issued as a few contiguous buckets in the order backward produces them (decoder
first), each launched the moment its last gradient has been written -- so all but the last bucket overlap with the
encoder backward.  RCCL runs on its own HIP stream; torch.distributed inserts the event fences against the compute
stream.  The 1/world factor is folded into the Adam kernel.  BatchNorm statistics stay per-replica, exactly like the
reference's DataParallel (no SyncBN).  Device-agnostic on purpose: the same code runs over gloo on CPU in the tests.
"""
import os

import torch
import torch.distributed as dist

# Optional record of every collective this module issues, in issue order: ("bucket", lo, hi) for a gradient bucket, ("stats", rows,
# cols, "sum" | "max") for a loss-statistics exchange.  The gradient buckets and the whole-batch loss statistics may travel on two
# different communicators (the own RCCL one and torch.distributed's); that is only safe when every rank issues them in the same order,
# which tests/test_distributed_cpu.py checks by comparing the ranks' logs.  None = off.
ISSUE_LOG = None


def _log(*what):
    if ISSUE_LOG is not None:
        ISSUE_LOG.append(tuple(what))


def agree_all_ranks(ok, process_group=None, device=None):
    """True iff `ok` is true on EVERY rank (MIN all-reduce of a flag over torch.distributed); with one rank: bool(ok).  Used to take
    collective decisions -- which data path the gradient exchange uses -- so that no subset of ranks ever ends up on a different
    communicator than its peers (that would hang the job without an error)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) <= 1:
        return bool(ok)
    backend = dist.get_backend(process_group)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if (backend == "nccl" and device is not None) else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=process_group)
    return bool(int(flag.item()) == 1)


def open_rccl_communicator(device, strict=False):
    """This library's own RCCL communicator on ALL ranks, or None on ALL ranks.  Two agreement rounds: (1) librccl loads everywhere --
    checked BEFORE rank 0's unique id is broadcast, so a rank that cannot load the library never leaves its peers blocked in the
    broadcast; (2) ncclCommInitRank succeeded everywhere -- otherwise every rank that did get a communicator destroys it and all of
    them use torch.distributed.  `strict` (an explicit request for the own path) turns a refusal into an error, also collectively."""
    import sys
    from . import rccl
    err = None
    try:
        rccl.load()
    except Exception as e:                       # noqa: BLE001
        err = e
    if not agree_all_ranks(err is None, device=device):
        if strict:
            raise RuntimeError("own RCCL communicator requested but librccl does not load on every rank (%s)" % (err,))
        if err is not None:
            print("supervised_dispnet_amd: librccl unavailable on this rank (%s); all ranks use torch.distributed" % err, file=sys.stderr)
        return None
    comm = None
    try:
        comm = rccl.Communicator(device=device)
    except Exception as e:                       # noqa: BLE001
        err = e
    if not agree_all_ranks(comm is not None, device=device):
        if comm is not None:
            comm.destroy()
        if strict:
            raise RuntimeError("own RCCL communicator requested but ncclCommInitRank did not succeed on every rank (%s)" % (err,))
        print("supervised_dispnet_amd: own RCCL communicator not available on every rank (%s); all ranks use torch.distributed" % (err,),
              file=sys.stderr)
        return None
    if comm.world > 1:
        # Preflight (round 6: the own communicator is the DEFAULT beyond one rank): one 1 MB sum through it against the same sum through
        # torch.distributed, waited for with a deadline -- a communicator that does not answer, or answers differently, is dropped on ALL
        # ranks before a gradient ever depends on it.
        why = _preflight(comm, device)
        if not agree_all_ranks(why is None, device=device):
            if strict:
                raise RuntimeError("own RCCL communicator failed its preflight on some rank (%s)" % (why,))
            print("supervised_dispnet_amd: own RCCL communicator failed its preflight (%s); all ranks use torch.distributed" % (why,), file=sys.stderr)
            if why is None or "deadline" not in why:
                comm.destroy()                # (one that never answered is left alone: destroying it would wait for it)
            return None
    return comm


def _preflight(comm, device, seconds=30.0):
    """None if a 1 MB fp32 sum over the ranks through `comm` completes within `seconds` and equals torch.distributed's (to summation
    order); otherwise the reason."""
    import time
    try:
        n = (1 << 20) // 4
        g = torch.Generator().manual_seed(4321 + comm.rank)
        src = torch.randn(n, generator=g).to(device)
        a, b = src.clone(), src.clone()
        comm.all_reduce_sum_(a, [torch.cuda.current_stream(device)])
        t0 = time.perf_counter()
        while not comm.stream.query():
            if time.perf_counter() - t0 > seconds:
                return "no answer within the %.0f s deadline" % seconds
            time.sleep(0.002)
        dist.all_reduce(b, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize(device)
        if not torch.allclose(a, b, rtol=1e-5, atol=1e-5):
            return "sum differs from torch.distributed's: max |diff| %.3g" % float((a - b).abs().max())
        return None
    except Exception as e:                       # noqa: BLE001
        return "%s: %s" % (type(e).__name__, str(e)[:160])


class GradReducer(object):
    """`comm`: "rccl" = this library's own communicator and HIP stream (rccl.Communicator: ncclCommInitRank / ncclAllReduce + event
    fences, the exchange launched as buckets complete, no host fence in between) -- the DEFAULT for device arenas at every world size
    since round 6: at world == 1 it is exercised bit for bit by tests/test_gpu_rccl.py; beyond one rank open_rccl_communicator takes it
    only if it initialises on EVERY rank and passes a preflight there (a 1 MB sum against torch.distributed's, with a deadline), and
    bench.py --gpus N compares and times the two paths once more after the timed region (config.rccl_selfcheck).  "torch" =
    torch.distributed.all_reduce(async_op=True) on the launcher's process group (backend "nccl" IS RCCL on ROCm; gloo on CPU in the
    tests): DN_COMM=torch, CPU arenas, a non-default process group, or the silent fall-back when the own communicator is refused.
    Or a ready rccl.Communicator.  Whatever is chosen is chosen by ALL ranks together.  `self.path` names the data path that actually
    runs ("rccl-own" / "torch.distributed:<backend>" / "none")."""

    def __init__(self, arena, bucket_bytes=20 << 20, process_group=None, comm=None, tail_bytes=1 << 20):
        self.arena = arena
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        import os
        env = os.environ.get("DN_COMM")
        default = "rccl" if (arena.flat_g.is_cuda and process_group is None) else "torch"
        choice = comm if comm is not None else (env or default)
        self.comm = None
        if choice == "rccl":
            if not arena.flat_g.is_cuda or process_group is not None:
                raise ValueError("the RCCL communicator serves device arenas on the default process group")
            self.comm = open_rccl_communicator(arena.flat_g.device, strict=(comm == "rccl" or env == "rccl") and self.world > 1)
            if self.comm is not None:
                self.world = self.comm.world
        elif choice != "torch":
            self.comm = choice
            self.world = choice.world
        if self.comm is not None:
            self.path = "rccl-own"
        elif self.world > 1:
            self.path = "torch.distributed:%s" % dist.get_backend(process_group)
        else:
            self.path = "none"
        # contiguous buckets over the arena (arena order == gradient production order).  The LAST bucket cannot overlap with anything
        # -- its all-reduce starts when the first layer's gradient lands and the optimizer waits for it -- so it is cut short: the
        # final `tail_bytes` of the arena (the first encoder layers, whose weights are small) instead of whatever the greedy split
        # leaves over (up to a whole bucket: ~20 MB = ~0.1 ms of exposed ring time on xGMI against ~1 MB)
        n = len(arena.params)
        tail_start, acc = n, 0
        if tail_bytes and tail_bytes > 0 and n > 1:
            for i in range(n - 1, 0, -1):
                acc += arena.params[i].numel() * 4
                tail_start = i
                if acc >= tail_bytes:
                    break
        self.buckets, start, acc = [], 0, 0
        for i, (p, o) in enumerate(zip(arena.params, arena.offsets)):
            acc += p.numel() * 4
            last = i == n - 1
            if acc >= bucket_bytes or last or i + 1 == tail_start:
                end = arena.numel if last else arena.offsets[i + 1]
                self.buckets.append({"lo": arena.offsets[start], "hi": end, "params": arena.params[start:i + 1]})
                start, acc = i + 1, 0
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._bucket_of[id(p)] = bi
        self._pending = None
        self._handles = []
        self.optimizer = None           # FusedAdam.overlap_backward(): ranges of the update are applied as the buckets complete
        self._opt_stream = None
        self._applied = set()
        self._lagging = None
        self._opt_used = set()
        self._step_applied = False
        self.reset()

    def reset(self):
        self._pending = [len(b["params"]) for b in self.buckets]
        self._handles = []

    def take_step_applied(self):
        """True once per finished backward pass whose buckets were all updated (consumed by FusedAdam.step)."""
        done, self._step_applied = self._step_applied, False
        return done

    def _host_work(self):
        """True when handing a bucket over involves work outside libdispnet_hip (a collective): under a launch tape such a call is a
        cut with a live host call; with one rank and no communicator everything a bucket triggers is launches and fences of this
        library and goes on the tape."""
        return self.world > 1 or self.comm is not None

    def _apply(self, bi, handle):
        """The optimizer's update of bucket bi, on its own stream, once the bucket's gradients are final (optimizer.overlap_backward)."""
        from . import engine
        b = self.buckets[bi]
        # Which stream: the weight-gradient side stream this event comes from (or the first one).  A stream of the optimizer's own
        # was measured SLOWER (b4 4.01 -> 4.18 ms, b32 +0.07 ms): a fourth concurrent queue costs more than the update's overlap gains,
        # like a third weight-gradient stream (engine.WGRAD_STREAMS); the dedicated stream is only the fallback without side streams.
        cur = torch.cuda.current_stream()
        sides = engine.side_stream()["sides"] if engine.wgrad_stream_enabled() else []
        if not sides:
            if self._opt_stream is None:
                self._opt_stream = torch.cuda.Stream(device=self.arena.flat_g.device)
            O = self._opt_stream
        else:
            O = cur if cur in sides else sides[0]
        self._opt_used.add(O)
        if self.comm is not None:
            engine.stream_wait(O, self.comm.stream)                  # behind the ncclAllReduce enqueued on the library's stream
        elif handle is not None:
            with torch.cuda.stream(O):
                handle.wait()                                        # O waits for this bucket's collective
        # ... and for everything enqueued so far on the compute streams: the gradients of the bucket (one rank: nothing else orders
        # them) and every kernel that still READS the bucket's parameters (see _bucket_ready).  Only O waits; no compute stream stalls.
        for st in engine.compute_streams():
            if st != O:
                engine.stream_wait(O, st)
        with engine.stream_scope(O):
            self.optimizer.apply_range(b["lo"], b["hi"], 1.0 / self.world, tick=not self._applied)
        self._applied.add(bi)

    def _bucket_ready(self, bi, handle):
        """Bucket bi is complete (and its collective enqueued).  Its update is applied ONE EVENT LATE -- when the next bucket completes,
        or in finish(): the layer whose weight gradient completed the bucket enqueues its input gradient right after, and that call
        may still re-lay the layer's weights from the arena (engine.ConvLayer.packed: first step, a new geometry); by the next event
        that read is enqueued, and the optimizer's stream waits for it."""
        if self._lagging is not None:
            self._apply(*self._lagging)
        self._lagging = (bi, handle)

    def _launch(self, bi):
        """Hand bucket bi to the communicator -- host work in the middle of the backward pass: under a launch tape
        (graph.TapedStep) the tape is cut here and this runs live between the replayed segments."""
        if self.arena.flat_g.is_cuda and self._host_work():
            from . import engine
            return engine.tape_host_call(lambda: self._launch_now(bi))
        return self._launch_now(bi)

    def _launch_now(self, bi):
        b = self.buckets[bi]
        _log("bucket", b["lo"], b["hi"])
        overlap = self.optimizer is not None and self.arena.flat_g.is_cuda
        if self.comm is not None:
            from . import engine                 # gradients of one bucket come from two HIP streams (engine.WGRAD_STREAM)
            self.comm.all_reduce_sum_(self.arena.flat_g[b["lo"]:b["hi"]], engine.compute_streams())
            self._used_comm = True
            if overlap:
                self._bucket_ready(bi, None)
            return
        handle = None
        if self.world > 1:
            if self.arena.flat_g.is_cuda:
                from . import engine
                engine.fence_streams()
            handle = dist.all_reduce(self.arena.flat_g[b["lo"]:b["hi"]], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._handles.append(handle)
        if overlap:
            self._bucket_ready(bi, handle)

    def grad_ready(self, param):
        """Engine hook: the gradient of `param` has been written into its arena view."""
        bi = self._bucket_of.get(id(param))
        if bi is None:
            return
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def finish(self):
        """Launch whatever did not complete through grad_ready (e.g. parameters without gradient this step), wait for
        all buckets (stream-level wait, no host sync on the nccl backend) and return the 1/world factor for the optimizer."""
        leftover = [bi for bi, n in enumerate(self._pending) if n > 0]       # (decided once: a tape replay repeats THIS step's decisions)
        if self.arena.flat_g.is_cuda and self._host_work():
            from . import engine
            engine.tape_host_call(lambda: self._finish_now(leftover))
        else:
            self._finish_now(leftover)
        if self.optimizer is not None and self._opt_used:
            from . import engine
            cur = torch.cuda.current_stream()
            for st in self._opt_used:
                if st != cur:
                    engine.stream_wait(cur, st)                                   # (recorded on a launch tape: outside the host call)
            self._opt_used = set()
        return 1.0 / self.world

    def _finish_now(self, leftover):
        for bi in leftover:
            self._launch_now(bi)
        for h in self._handles:
            h.wait()
        if self.comm is not None:
            self.comm.join()                     # the optimizer (current stream) waits for the last bucket; no host sync
        if self.optimizer is not None and self.arena.flat_g.is_cuda:
            if self._lagging is not None:
                self._apply(*self._lagging)
                self._lagging = None
            if len(self._applied) != len(self.buckets):
                raise RuntimeError("GradReducer: %d of %d buckets updated" % (len(self._applied), len(self.buckets)))
            self._applied = set()                # (here, not in the optimizer: a launch tape replays this call, not optimizer.step())
            self._step_applied = True
        self.reset()


def shard_slice(global_batch, rank, world):
    """Contiguous per-rank slice of a global batch, in rank order -- the split nn.DataParallel.scatter makes
    (train.py:316); the global batch must divide evenly so every rank steps the same shapes."""
    if global_batch % world != 0:
        raise ValueError("global batch %d does not divide over %d ranks" % (global_batch, world))
    per = global_batch // world
    return slice(rank * per, (rank + 1) * per)


def data_parallel_world(process_group=None):
    """World size of the data-parallel job (1 when torch.distributed is not initialised)."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(process_group)
    return 1


def exchange_loss_stats(stats, sum_cols, max_cols=(), process_group=None):
    """In-place exchange of per-group loss statistics [G, S] between the ranks: columns `sum_cols` are summed, columns
    `max_cols` take the maximum, the others are left alone.  This is the (sum, count) exchange of SURVEY.md 8e that makes the
    Multiscale_* / DORN whole-batch normalisation (loss_functions.py:232-237, 72-73) identical to the reference's, which sees the
    gathered batch on GPU0: a handful of floats per step."""
    if data_parallel_world(process_group) <= 1:
        return stats
    if stats.is_cuda:
        from . import engine
        engine._not_on_tape("the whole-batch loss statistics exchange (framework-side slicing + all-reduce in the forward pass)")
    if sum_cols:
        idx = list(sum_cols)
        part = stats[:, idx].contiguous()
        _log("stats", part.shape[0], part.shape[1], "sum")
        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=process_group)
        stats[:, idx] = part
    if max_cols:
        idx = list(max_cols)
        part = stats[:, idx].contiguous()
        _log("stats", part.shape[0], part.shape[1], "max")
        dist.all_reduce(part, op=dist.ReduceOp.MAX, group=process_group)
        stats[:, idx] = part
    return stats


def global_masked_mean(local_sum, local_count, process_group=None):
    """sum / count over ALL ranks.  The Multiscale_* losses and DORN_loss normalise by the valid-pixel count of the whole
    batch (loss_functions.py:232-237, 72-73), which the reference's DataParallel sees gathered on GPU0; under one process
    per GPU the (sum, count) pairs are exchanged (one tiny all-reduce) before the division so the value is identical."""
    pair = torch.stack([local_sum.reshape(-1), local_count.reshape(-1).to(local_sum.dtype)])
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.all_reduce(pair, op=dist.ReduceOp.SUM, group=process_group)
    return pair[0] / pair[1]


def average_plain_grads(params, process_group=None):
    """Data-parallel exchange for the torch.optim optimizers (--sgd / --diff-lr, train.py:306-314): the `.grad` tensors are
    summed over the ranks in ONE flattened all-reduce and divided by the world size, which is what nn.DataParallel's gradient
    reduce gives the reference for every optimizer.  Parameters without a gradient on this step contribute zeros (every rank
    walks the same list, so the collective shapes agree)."""
    world = data_parallel_world(process_group)
    if world <= 1:
        return
    params = list(params)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=process_group)
    flat.div_(world)
    off = 0
    for p in params:
        n = p.numel()
        if p.grad is not None:
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n
