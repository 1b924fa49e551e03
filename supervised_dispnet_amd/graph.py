"""Whole-step hipGraph capture: forward + loss + backward + optimizer as ONE graph launch.

A Disp_vgg_BN training step is ~330 kernel launches driven from Python (ctypes) at ~18 us each: ~6 ms of host time.  At the
metric's b32 per GPU the device needs 23 ms and the host runs ahead; at 32/8 = 4 images per GPU (BASELINE.json's "b32 @ 8 GPU")
the device needs ~4 ms and the host is the bottleneck.  The launch sequence of a step is static (same shapes, same kernels, the
weight re-lays included), so it is captured once with HIP stream capture (torch.cuda.CUDAGraph: the caching allocator gives the
capture a private pool, so every activation / workspace / packed-weight pointer baked into the graph stays valid) and replayed
with one hipGraphLaunch per step.

Requirements on the step callable (all met by the engine):
  * no host synchronisation and no data-dependent host control flow (the engine has none; `.item()` on the loss is the caller's,
    after the replay);
  * the optimizer keeps its step counter on the device (FusedAdam.capturable());
  * the side stream of the weight gradients forks from and joins back into the capture stream (engine.join_side_stream);
  * inputs live in fixed buffers: `GraphedStep.inputs` are the tensors the step closes over -- write the next batch into them
    (copy_) before calling the graph.
Multi-rank: the gradient all-reduce (RCCL via torch.distributed) is capturable by ProcessGroupNCCL but that path cannot be
exercised on this one-GPU pool, so callers decide (bench.py: --graph auto = single-rank only).
"""
import os
import time

import torch

from . import engine


class GraphedStep(object):
    def __init__(self, step, optimizer=None, warmup=3, static_inputs=()):
        """`step()` runs one whole iteration and returns a tensor (the loss) or a tuple of tensors.  `warmup` eager iterations
        run on the capture stream first (allocator warm-up, lazy kernel attribute setup, packed-weight caches)."""
        self.step = step
        self.inputs = tuple(static_inputs)
        self.graph = None
        self.out = None
        if optimizer is not None and hasattr(optimizer, "capturable"):
            optimizer.capturable(True)
        self._stream = torch.cuda.Stream()
        self._warmup = warmup

    def capture(self):
        s = self._stream
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self._warmup):
                self.step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        engine.CAPTURING = True
        try:
            with torch.cuda.graph(self.graph, stream=s):
                self.out = self.step()
        finally:
            engine.CAPTURING = False
        return self

    def __call__(self):
        if self.graph is None:
            self.capture()
        self.graph.replay()
        return self.out


_ONES = {}


def backward(loss):
    """loss.backward() seeded from a persistent ones tensor: autograd's own seed is a fresh fill launched by the framework, which a
    launch tape cannot see (a step handed to TapedStep must use this)."""
    key = (loss.device, loss.dtype, tuple(loss.shape))
    one = _ONES.get(key)
    if one is None:
        with engine.outside_tape_pool():      # (persistent + initialised by the framework: must not sit in a tape's private pool)
            one = _ONES[key] = torch.ones_like(loss)
    loss.backward(one)


class TapedStep(object):
    """Whole-step launch tape: forward + loss + backward + optimizer recorded ONCE as the list of this library's kernel launches and
    stream fences (dn_tape_*, csrc/dn_tape.hip) and re-issued by one C call per step -- the device sees exactly the eager launch
    sequence (two compute streams, same fences) while the host spends ~2 us per launch instead of ~20.  At the metric's 32 / 8 = 4
    images per GPU the Python side of a step costs as much as the device side (tools/host_overhead.py); a hipGraph of the same step
    replays slower on the device than the eager launches.

    The recorded step is a REAL step (it runs eagerly and counts).  Requirements on `step` beyond GraphedStep's (no host
    synchronisation, device-side optimizer counter, inputs in fixed buffers):
      * every piece of device work is a launch of libdispnet_hip (no framework fills / copies / elementwise kernels: the engine
        refuses the ones it knows about while a tape is recorded; seed the backward with graph.backward(loss));
      * tensors are allocated under a private pool of the caching allocator while recording, so every pointer the tape holds stays
        reserved; the result tensors `step` returns are the tape's static outputs.
    Host work between launches that must ALSO happen at replay (a gradient bucket's all-reduce on torch.distributed) is registered
    with engine.tape_host_call(fn): the tape is cut there and `fn` runs live between the segments, on the tape's stream."""

    def __init__(self, step, optimizer=None, warmup=3, static_inputs=(), lazy_join=False):
        self.step = step
        # lazy_join: a replay does NOT make the caller's stream wait for the tape's stream; the caller calls join() before it reads the
        # outputs (or anything else the step wrote) on its own stream.  Back-to-back replays then follow each other on the tape's stream
        # without the round trip through the caller's stream (record on the tape's stream -> wait on the caller's -> record there -> wait
        # on the tape's: ~25 us of idle device between two steps, tools/step_gaps.py).
        self.lazy_join = lazy_join
        self.inputs = tuple(static_inputs)
        self.out = None
        self.tape = None
        self.host_calls = []
        self.host_s, self.replays = 0.0, 0          # host time spent re-issuing, replays so far
        if optimizer is not None and hasattr(optimizer, "capturable"):
            optimizer.capturable(True)
        self._stream = torch.cuda.Stream()
        self._warmup = warmup
        self._pool = None
        self._keep = None
        self._pid = os.getpid()

    def capture(self):
        from . import _lib
        lib = _lib.load()
        s = self._stream
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self._warmup):
                self.step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        dev = torch.cuda.current_device()
        # models of earlier runs that are garbage but not collected yet still have rows in the batched weight re-lay: drop them before
        # the recording takes a snapshot of that table (it keeps what it snapshots alive, see engine.PackTable.run)
        import gc
        gc.collect()
        for t in engine._PACK_TABLES.values():
            t._prune()
        if engine.SPLITK:
            engine.xcd_placement_ok(torch.device("cuda", dev))     # the one-time placement probe belongs in front of the recording
        self._pool = torch.cuda.MemPool()
        handle = lib.dn_tape_begin()
        if not handle:
            raise _lib.DispnetHipError("dn_tape_begin: " + _lib.last_error())
        rec = {"handle": handle, "keep": [], "host_calls": self.host_calls, "paused": False, "pool": (dev, self._pool.id)}
        engine.TAPE = rec
        torch._C._cuda_beginAllocateToPool(dev, self._pool.id)          # every thread (the backward runs on autograd's)
        try:
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self.out = self.step()
            torch.cuda.current_stream().wait_stream(s)
        except BaseException:
            torch._C._cuda_endAllocateToPool(dev, self._pool.id)
            torch._C._cuda_releasePool(dev, self._pool.id)
            engine.TAPE = None
            lib.dn_tape_free(handle)          # (also ends the recording)
            del self.host_calls[:]
            raise
        torch._C._cuda_endAllocateToPool(dev, self._pool.id)
        torch._C._cuda_releasePool(dev, self._pool.id)       # (the begin's reference; the MemPool object keeps the pool -- and every
        engine.TAPE = None                                    #  block the tape points into -- alive for as long as this TapedStep lives)
        _lib.call("dn_tape_end", handle)
        torch.cuda.synchronize()
        self._keep = rec["keep"]          # (cheap to hold; dropping them would only return the blocks to the private pool)
        self.tape = handle
        self._lib = lib
        self.segments = lib.dn_tape_segments(handle)
        self.launches, self.fences = lib.dn_tape_launches(handle), lib.dn_tape_fences(handle)
        self.riding_fences = lib.dn_tape_riding_fences(handle)      # fences re-issued as the stop event of the launch in front of them
        if self.segments != len(self.host_calls) + 1:
            raise RuntimeError("launch tape: %d segments for %d host calls" % (self.segments, len(self.host_calls)))
        return self

    def __call__(self):
        if self.tape is None:
            self.capture()
            return self.out
        s = self._stream
        cur = torch.cuda.current_stream()
        if not (self.lazy_join and cur.query()):
            s.wait_stream(cur)            # the caller's writes into the static inputs (skipped when the caller's stream is idle: nothing pending there can
                                          # still be writing them -- with lazy_join that stream does not even hold the previous replay's join)
        t_host = time.perf_counter()
        if not self.host_calls:
            rc = self._lib.dn_tape_replay(self.tape, -1)
        else:
            rc = 0
            for i in range(self.segments):
                rc = self._lib.dn_tape_replay(self.tape, i)
                if rc != 0:               # a failed segment: no further collectives on top of it
                    break
                if i < len(self.host_calls):
                    fn, st = self.host_calls[i]
                    with torch.cuda.stream(st), engine.stream_scope():      # the stream the call was issued on when recorded
                        fn()
        if rc != 0:
            from . import _lib
            raise _lib.DispnetHipError("dn_tape_replay failed (%d): %s" % (rc, _lib.last_error()))
        self.host_s += time.perf_counter() - t_host     # host time of the re-issue (bench.py reports it per step)
        self.replays += 1
        if not self.lazy_join:
            cur.wait_stream(s)
        engine.bump_param_epoch()         # the replayed optimizer changed the weights: eager code that follows re-lays its packed copies
        return self.out

    def join(self):
        """The caller's current stream waits for everything replayed so far (needed with lazy_join before the outputs are read there,
        and before the static inputs are overwritten from there)."""
        if self.tape is not None:
            torch.cuda.current_stream().wait_stream(self._stream)

    def write_inputs(self, *srcs):
        """Refill the static inputs for the next replay: static_inputs[i].copy_(srcs[i]) ON THE TAPE'S STREAM, after whatever the caller's
        stream has pending for `srcs`.  Stream order puts the copies behind the previous replay's reads of the inputs and in front of
        the next replay's -- safe with lazy_join without a join (a copy_ issued on the caller's stream would race with the tail of the
        previous replay: write after read)."""
        if len(srcs) != len(self.inputs):
            raise ValueError("write_inputs: %d tensors for %d static inputs" % (len(srcs), len(self.inputs)))
        s = self._stream
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            for dst, src in zip(self.inputs, srcs):
                dst.copy_(src, non_blocking=True)
                if src.is_cuda:
                    src.record_stream(s)

    def verify(self, state):
        """Replay once and run the SAME step eagerly once from the same state (`state`: the tensors the step reads and updates --
        parameters, optimizer moments and counters, BatchNorm buffers), and compare the resulting state and outputs bit for bit.
        Leaves the state one step further (the eager result).  Returns (identical, largest absolute difference).  What the unit tests
        pin at small sizes, checked on the caller's own workload and size (bench.py reports it in config.tape_verified).
        A caller whose inputs change from step to step writes a batch DIFFERENT from the recorded one into the static inputs first:
        framework-side work the tape missed then shows up as stale data (train.py --tape does)."""
        if self.tape is None:
            self.capture()
        torch.cuda.synchronize()
        before = [t.clone() for t in state]
        out = self()
        self.join()
        outs_r = [o.clone() for o in (out if isinstance(out, (tuple, list)) else (out,))]
        torch.cuda.synchronize()
        after_r = [t.clone() for t in state]
        with torch.no_grad():
            for t, b in zip(state, before):
                t.copy_(b)
        s = self._stream
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            out_e = self.step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        outs_e = list(out_e if isinstance(out_e, (tuple, list)) else (out_e,))
        self.last_eager_out = outs_e
        worst, same = 0.0, True
        for a, b in list(zip(after_r, state)) + list(zip(outs_r, outs_e)):
            if not torch.equal(a, b):
                same = False
                worst = max(worst, float((a.double() - b.double()).abs().max()))
        return same, worst

    def close(self):
        """Free the tape and the private pool now (idempotent).  Call it when the taped training is over rather than leaving it to the
        garbage collector: the step closure usually refers back to its owner (a cycle), and a process forked in the meantime -- a
        DataLoader worker -- would inherit the garbage and run this destructor without the HIP context behind it."""
        if self._pid != os.getpid():
            return
        if self.tape is not None:
            self._lib.dn_tape_free(self.tape)
            self.tape = None
        self.host_calls = []
        self.step = None
        self._keep = None
        self.out = None
        self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:                 # noqa: BLE001  (interpreter shutdown)
            pass
