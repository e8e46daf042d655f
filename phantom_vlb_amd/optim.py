"""AdamW on libvlb (vlb_adamw_step) behind the torch.optim.Optimizer interface.

Mirrors ``torch.optim.AdamW(lr, betas, eps, weight_decay)`` as the reference configures it
(src/litmodule/videollama2_vlb_litmodule.py:357-363) so ``CosineAnnealingLR`` and a Lightning
Trainer can drive it unchanged.  Differences, documented in DESIGN.md: the update runs on fp32
masters with fp32 moments (the reference updates bf16 params in place), the global-norm clip of
``Trainer(gradient_clip_val=1)`` is fused into the same kernel (device-side norm, no host sync),
and the bf16 compute copies read by the kernels are refreshed in the same launch.

All state lives in the flat buffers of ``flat.FlatTrainables``: one sum-of-squares launch and one
clip+AdamW launch per step.  Under data parallelism ``parallel.ShardedFlatState`` is attached and
the same two launches run on this rank's 1/world slice between a gradient reduce-scatter and an
all-gather of the refreshed bf16 copies.
"""
from __future__ import annotations

import torch

from ._lib import check, lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


class VlbAdamW(torch.optim.Optimizer):
    def __init__(self, named_params, flat, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 max_norm: float = 0.0):
        named_params = list(named_params)
        self.names = [n for n, _ in named_params]
        params = [p for _, p in named_params]
        for p in params:
            assert p.dtype == torch.float32 and p.is_cuda and p.is_contiguous()
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_norm = float(max_norm)
        self.flat = flat                  # FlatTrainables: masters / bf16 copies / grads / moments
        self.sharded = None               # parallel.ShardedFlatState under data parallelism (1/world of every buffer)
        dev = params[0].device
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.sumsq_ws = torch.zeros(lib.vlb_sumsq_ws_floats(), dtype=torch.float32, device=dev)
        self.step_count = 0
        self.post_step = []           # callables run after every update (e.g. LoRA derived layouts)

    def attach_sharded(self, state):
        """Data parallelism: gradients are reduce-scattered, this rank updates its 1/world slice of every segment
        and the bf16 copies are all-gathered (parallel.ShardedFlatState)."""
        assert state.flat is self.flat
        self.sharded = state

    def full_state(self, name: str):
        """Full-size flat fp32 buffer 'master' | 'm' | 'v' (gathered from the shards under data parallelism)."""
        return self.sharded.gather_full(name) if self.sharded is not None else getattr(self.flat, name)

    def load_full_state(self, name: str, full):
        full = full.to(self.flat.master.device)
        if self.sharded is not None:
            self.sharded.load_full(name, full)
            if name == "master":
                self.flat.master.copy_(full)
        else:
            getattr(self.flat, name).copy_(full)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        group = self.param_groups[0]
        self.step_count += 1
        st = _stream()
        sh = self.sharded
        if sh is not None:
            sh.finish_reduce()               # reduce-scatters started under backward + the rest (head)
        f = sh if sh is not None else self.flat
        self.sumsq.zero_()
        b1, b2 = group["betas"]
        if self.max_norm > 0:
            check(lib.vlb_grad_sumsq(f.grad.data_ptr(), f.numel, self.sumsq.data_ptr(), self.sumsq_ws.data_ptr(), st), "vlb_grad_sumsq")
            if sh is not None:
                sh.all_reduce_scalar(self.sumsq)      # the clip norm covers every rank's slices
        check(lib.vlb_adamw_step(f.master.data_ptr(), f.compute.data_ptr(), f.grad.data_ptr(), f.m.data_ptr(),
                                 f.v.data_ptr(), f.numel, float(group["lr"]), float(b1), float(b2),
                                 float(group["eps"]), float(group["weight_decay"]), self.step_count,
                                 self.sumsq.data_ptr(), self.max_norm, st), "vlb_adamw_step")
        if sh is not None:
            sh.gather_compute()
        for fn in self.post_step:
            fn()
        return loss

    def grad_norm(self) -> float:
        """Global gradient norm of the last step (host sync; for logging only)."""
        return float(self.sumsq.sqrt().item())
