"""AdamW on libvlb (vlb_adamw_step) behind the torch.optim.Optimizer interface.

Mirrors ``torch.optim.AdamW(lr, betas, eps, weight_decay)`` as the reference configures it
(src/litmodule/videollama2_vlb_litmodule.py:357-363) so ``CosineAnnealingLR`` and a Lightning
Trainer can drive it unchanged.  Differences, documented in DESIGN.md: the update runs on fp32
masters with fp32 moments (the reference updates bf16 params in place), the global-norm clip of
``Trainer(gradient_clip_val=1)`` is fused into the same kernel (device-side norm, no host sync),
and the bf16 compute copies read by the kernels are refreshed in the same launch.
"""
from __future__ import annotations

import torch

from ._lib import check, lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


class VlbAdamW(torch.optim.Optimizer):
    def __init__(self, named_params, bf16_copies: dict, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 max_norm: float = 0.0, flat=None):
        named_params = list(named_params)
        self.names = [n for n, _ in named_params]
        params = [p for _, p in named_params]
        for p in params:
            assert p.dtype == torch.float32 and p.is_cuda and p.is_contiguous()
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.bf16 = [bf16_copies.get(n) for n in self.names]
        self.max_norm = float(max_norm)
        self.flat = flat                  # FlatTrainables: one launch per step instead of one per tensor
        if flat is not None:
            self.m = [flat.m[o:o + k].view(shp) for (o, k, shp) in flat.offsets.values()]
            self.v = [flat.v[o:o + k].view(shp) for (o, k, shp) in flat.offsets.values()]
        else:
            self.m = [torch.zeros_like(p) for p in params]
            self.v = [torch.zeros_like(p) for p in params]
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=params[0].device)
        self.sumsq_ws = torch.zeros(lib.vlb_sumsq_ws_floats(), dtype=torch.float32, device=params[0].device)
        self.step_count = 0
        self.grad_reducer = None      # set by the data-parallel wrapper: callable(list_of_grads)
        self.post_step = []           # callables run after every update (e.g. LoRA derived layouts)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        group = self.param_groups[0]
        params = group["params"]
        grads = [p.grad for p in params]
        if self.grad_reducer is not None:
            self.grad_reducer(grads)
        self.step_count += 1
        st = _stream()
        self.sumsq.zero_()
        if self.flat is not None:
            f = self.flat
            b1, b2 = group["betas"]
            if self.max_norm > 0:
                check(lib.vlb_grad_sumsq(f.grad.data_ptr(), f.numel, self.sumsq.data_ptr(), self.sumsq_ws.data_ptr(), st), "vlb_grad_sumsq")
            check(lib.vlb_adamw_step(f.master.data_ptr(), f.compute.data_ptr(), f.grad.data_ptr(), f.m.data_ptr(),
                                     f.v.data_ptr(), f.numel, float(group["lr"]), float(b1), float(b2),
                                     float(group["eps"]), float(group["weight_decay"]), self.step_count,
                                     self.sumsq.data_ptr(), self.max_norm, st), "vlb_adamw_step")
            for fn in self.post_step:
                fn()
            return loss
        if self.max_norm > 0:
            for g in grads:
                check(lib.vlb_grad_sumsq(g.data_ptr(), g.numel(), self.sumsq.data_ptr(), self.sumsq_ws.data_ptr(), st), "vlb_grad_sumsq")
        b1, b2 = group["betas"]
        for p, g, m, v, pb in zip(params, grads, self.m, self.v, self.bf16):
            check(lib.vlb_adamw_step(p.data_ptr(), None if pb is None else pb.data_ptr(), g.data_ptr(), m.data_ptr(),
                                     v.data_ptr(), p.numel(), float(group["lr"]), float(b1), float(b2),
                                     float(group["eps"]), float(group["weight_decay"]), self.step_count,
                                     self.sumsq.data_ptr(), self.max_norm, st), "vlb_adamw_step")
        for fn in self.post_step:
            fn()
        return loss

    def grad_norm(self) -> float:
        """Global gradient norm of the last step (host sync; for logging only)."""
        return float(self.sumsq.sqrt().item())
