"""Fused Adam over a flat parameter arena (reference: torch.optim.Adam at train.py:303-305,520-522).

All hot parameters are re-pointed at slices of ONE fp32 buffer; gradients, exp_avg and exp_avg_sq live in three more.
The engine writes parameter gradients straight into the gradient arena (`param._dn_grad_view`), so a step is a single
`dn_adam_step` launch over ~20 M elements and the data-parallel all-reduce runs over the same flat buffer in buckets.
Numerics follow torch.optim.Adam (amsgrad off): lerp/addcmul updates, denom = sqrt(v)/sqrt(1-b2^t) + eps.
"""
import torch

from . import _lib, engine


class FusedAdam(object):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("FusedAdam got no trainable parameters")
        dev = params[0].device
        engine.require_cuda(params[0], "parameters")
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.params = params
        self.step_count = 0
        n = sum(p.numel() for p in params)
        # 4-element alignment of every slice keeps float4 loads legal inside the kernels that read parameters directly
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.numel, self.offsets = total, offs
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(params, offs):
                view = self.flat_p[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p._dn_grad_view = self.flat_g[o:o + p.numel()].view(p.shape)
        engine.bump_param_epoch()
        self.param_groups = [{"params": params, "lr": self.lr, "betas": self.betas, "eps": self.eps,
                              "weight_decay": self.weight_decay}]

    def zero_grad(self, set_to_none=True):
        # gradients are fully overwritten by the engine each backward; only stray autograd .grad tensors need clearing
        for p in self.params:
            p.grad = None

    def _gather_stray_grads(self):
        """Parameters whose gradient arrived through autograd (.grad) instead of the in-place sink."""
        for p in self.params:
            if p.grad is not None:
                p._dn_grad_view.copy_(p.grad)
                p.grad = None

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        self._gather_stray_grads()
        self.step_count += 1
        g = self.param_groups[0]
        _lib.call("dn_adam_step", self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(),
                  self.exp_avg_sq.data_ptr(), self.numel, float(g["lr"]), self.betas[0], self.betas[1], self.eps,
                  self.weight_decay, self.step_count, float(grad_scale), engine._stream())
        engine.bump_param_epoch()

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
