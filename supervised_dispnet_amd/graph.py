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
