"""Flat parameter arena + fused Adam (reference: torch.optim.Adam at train.py:303-305,520-522).

ParamArena (device-agnostic, pure plumbing): re-points every hot parameter at a slice of ONE fp32 buffer and gives each a
gradient view into a second one (`param._dn_grad_view`), ordered by the order gradients are PRODUCED in backward, so
the data-parallel reducer can all-reduce contiguous buckets as soon as they are complete.
FusedAdam (HIP): one `dn_adam_step` launch over the whole arena; numerics follow torch.optim.Adam (amsgrad off):
lerp / addcmul updates, denom = sqrt(v)/sqrt(1-b2^t) + eps, step_size = lr/(1-b1^t); grad * (1/world) folded in.
"""
import torch

from . import _lib, engine


class ParamArena(object):
    def __init__(self, params, production_order=None):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("ParamArena got no trainable parameters")
        if production_order is not None:
            rank = {id(p): i for i, p in enumerate(production_order)}
            params = sorted(params, key=lambda p: rank.get(id(p), len(rank)))
        self.params = params
        dev = params[0].device
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4        # 16-byte aligned slices
        self.numel, self.offsets = total, offs
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(params, offs):
                view = self.flat_p[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p._dn_grad_view = self.flat_g[o:o + p.numel()].view(p.shape)
        engine.bump_param_epoch()

    def gather_stray_grads(self):
        """Parameters whose gradient arrived through autograd (.grad) instead of the engine's in-place sink."""
        for p in self.params:
            if p.grad is not None:
                p._dn_grad_view.copy_(p.grad)
                p.grad = None


class FusedAdam(object):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, production_order=None):
        self.arena = params if isinstance(params, ParamArena) else ParamArena(list(params), production_order)
        engine.require_cuda(self.arena.flat_p, "parameters")
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.step_count = 0
        self.exp_avg = torch.zeros_like(self.arena.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.arena.flat_p)
        self.param_groups = [{"params": self.arena.params, "lr": self.lr, "betas": self.betas, "eps": self.eps,
                              "weight_decay": self.weight_decay}]
        self._dev = None          # capturable(): device-resident step counter / hyper-parameters for hipGraph replay
        self._overlap = None      # overlap_backward(): the bucket watcher that applies ranges of the update as their gradients complete

    def capturable(self, on=True):
        """Keep the step counter and (lr, beta1, beta2) on the DEVICE (dn_adam_step_dev) so that a captured hipGraph of the
        training step advances the bias corrections on every replay; a scheduler changes lr through set_lr()."""
        if on and self._dev is None:
            dev = self.arena.flat_p.device
            g = self.param_groups[0]
            self._dev = {"hyper": torch.tensor([float(g["lr"]), self.betas[0], self.betas[1]], dtype=torch.float64, device=dev),
                         "step": torch.tensor([self.step_count], dtype=torch.int32, device=dev),
                         "derived": torch.zeros(4, dtype=torch.float32, device=dev), "lr": float(g["lr"])}
        elif not on and self._dev is not None:
            self.step_count = int(self._dev["step"].item())
            self._dev = None
        return self

    def set_lr(self, lr):
        self.param_groups[0]["lr"] = float(lr)
        if self._dev is not None and self._dev["lr"] != float(lr):
            self._dev["hyper"][0:1].fill_(float(lr))
            self._dev["lr"] = float(lr)

    @property
    def params(self):
        return self.arena.params

    def zero_grad(self, set_to_none=True):
        # arena gradients are fully overwritten by the engine every backward; only stray autograd grads need clearing
        for p in self.arena.params:
            p.grad = None

    def overlap_backward(self, reducer):
        """Apply the update a gradient bucket at a time, as soon as the bucket's gradients (and, under data parallelism, their
        all-reduce) are complete, on a stream of its own under the rest of the backward pass (distributed.GradReducer drives it; with
        one rank the reducer exchanges nothing and only watches the buckets).  Element-wise arithmetic: bit-identical to one update
        of the whole arena.  What is left at step() is bookkeeping; the unoverlappable tail of a step shrinks from the whole
        arena's 0.09 ms pass to the last ~1 MB bucket's.  Contract: exactly one backward pass per step(), through the engine (no
        gradient accumulation, no stray autograd .grad on the arena's parameters)."""
        self._overlap = reducer
        if reducer is not None:
            reducer.optimizer = self
        return self

    @torch.no_grad()
    def apply_range(self, lo, hi, grad_scale=1.0, tick=True):
        """Adam update of arena elements [lo, hi) on the current stream (engine.stream_scope); `tick`: this is the first range of the
        optimizer step (the device-side counter and bias corrections advance once per step)."""
        a = self.arena
        g = self.param_groups[0]
        n = hi - lo
        ptr = lambda t: t.data_ptr() + 4 * lo
        if self._dev is not None:
            d = self._dev
            if d["lr"] != float(g["lr"]):
                self.set_lr(g["lr"])
            if tick:
                _lib.call("dn_adam_tick", d["hyper"].data_ptr(), d["step"].data_ptr(), d["derived"].data_ptr(), engine._stream())
            engine.hbm_call("dn::adam_dev_kernel", n * 28, "dn_adam_step_dev", ptr(a.flat_p), ptr(a.flat_g), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                            n, d["hyper"].data_ptr(), self.eps, self.weight_decay, None, d["derived"].data_ptr(), float(grad_scale), engine._stream())
        else:
            engine.hbm_call("dn::adam_kernel", n * 28, "dn_adam_step", ptr(a.flat_p), ptr(a.flat_g), ptr(self.exp_avg), ptr(self.exp_avg_sq), n,
                            float(g["lr"]), self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count + 1, float(grad_scale),
                            engine._stream())

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        a = self.arena
        if self._overlap is not None:
            # every range was applied as its bucket completed (GradReducer.finish() has applied the stragglers and joined the stream)
            if not self._overlap.take_step_applied():
                raise RuntimeError("FusedAdam.overlap_backward: step() without a backward pass + GradReducer.finish() before it")
            if any(p.grad is not None for p in a.params):
                raise RuntimeError("FusedAdam.overlap_backward: a parameter received its gradient through autograd (.grad), not the engine")
            self.step_count += 1
            engine.bump_param_epoch()
            return
        a.gather_stray_grads()
        self.step_count += 1
        g = self.param_groups[0]
        if self._dev is not None:
            d = self._dev
            if d["lr"] != float(g["lr"]):
                self.set_lr(g["lr"])
            engine.hbm_call("dn::adam_dev_kernel", a.numel * 28, "dn_adam_step_dev", a.flat_p.data_ptr(), a.flat_g.data_ptr(),
                            self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), a.numel, d["hyper"].data_ptr(), self.eps,
                            self.weight_decay, d["step"].data_ptr(), d["derived"].data_ptr(), float(grad_scale), engine._stream())
            engine.bump_param_epoch()
            return
        engine.hbm_call("dn::adam_kernel", a.numel * 28, "dn_adam_step", a.flat_p.data_ptr(), a.flat_g.data_ptr(), self.exp_avg.data_ptr(),
                  self.exp_avg_sq.data_ptr(), a.numel, float(g["lr"]), self.betas[0], self.betas[1], self.eps,
                  self.weight_decay, self.step_count, float(grad_scale), engine._stream())
        engine.bump_param_epoch()

    def state_dict(self):
        if self._dev is not None:
            self.step_count = int(self._dev["step"].item())
        return {"step": self.step_count, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        if self._dev is not None:
            self._dev["step"].fill_(self.step_count)
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
