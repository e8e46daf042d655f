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
    """``flat``: one flat store or a list of them (head + LoRA: fp32 gradients; ``fullft.FlatBackbone``: bf16
    gradients, attribute ``grad_bf16``).  One clip norm over all of them, one AdamW launch per store."""

    def __init__(self, named_params, flat, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 max_norm: float = 0.0):
        named_params = list(named_params)
        self.names = [n for n, _ in named_params]
        params = [p for _, p in named_params]
        for p in params:
            assert p.dtype == torch.float32 and p.is_cuda and p.is_contiguous()
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_norm = float(max_norm)
        self.flats = list(flat) if isinstance(flat, (list, tuple)) else [flat]
        self.flat = self.flats[0]         # head (+ LoRA) store
        self.shardeds = [None] * len(self.flats)      # parallel.ShardedFlatState per store under data parallelism
        dev = params[0].device
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.sumsq_ws = torch.zeros(max(lib.vlb_sumsq_ws_floats(), 1024), dtype=torch.float32, device=dev)
        self.step_count = 0
        self.post_step = []           # callables run after every update (e.g. LoRA derived layouts, W^T refresh)

    @property
    def sharded(self):
        return self.shardeds[0]

    def attach_sharded(self, state):
        """Data parallelism: gradients are reduce-scattered, this rank updates its 1/world slice of every segment
        and the bf16 copies are all-gathered (parallel.ShardedFlatState); one state per flat store."""
        i = [k for k, f in enumerate(self.flats) if f is state.flat]
        assert i, "attach_sharded: the state wraps a flat store this optimiser does not own"
        self.shardeds[i[0]] = state

    def full_state(self, name: str, index: int = 0):
        """Full-size flat fp32 buffer 'master' | 'm' | 'v' of store `index` (gathered from the shards under data parallelism)."""
        sh = self.shardeds[index]
        return sh.gather_full(name) if sh is not None else getattr(self.flats[index], name)

    def load_full_state(self, name: str, full, index: int = 0):
        f, sh = self.flats[index], self.shardeds[index]
        full = full.to(f.compute.device)
        if sh is not None:
            sh.load_full(name, full)
            if name == "master" and f.master is not None:       # (None: FULL_SHARD keeps no full-size staging copy)
                f.master.copy_(full)
        else:
            getattr(f, name).copy_(full)

    def compute_from_master(self, index: int):
        """bf16 weights of store `index` <- its restored fp32 masters (checkpoint resume)."""
        f, sh = self.flats[index], self.shardeds[index]
        if sh is not None:
            sh.compute_from_master()
        else:
            chunk = 1 << 28
            for a in range(0, f.numel, chunk):
                f.compute[a:a + chunk].copy_(f.master[a:a + chunk])

    # ------------------------------------------------------------------ checkpoint surface (torch.optim.Optimizer)
    def state_dict(self):
        """``torch.optim`` layout (``state`` / ``param_groups``) plus ``vlb``: the bias-correction step count and, per flat
        store, both fp32 Adam moments (gathered under data parallelism) - and for the stores whose weights are NOT in the
        module's ``state_dict()`` (index >= 1: the full fine-tune's backbone in kernel layouts) the fp32 masters too.  This
        is what a Lightning ``ModelCheckpoint`` stores under ``optimizer_states``; the built-in runner's own checkpoint
        (trainer.trainable_state) carries the same tensors."""
        sd = super().state_dict()
        stores = []
        for i in range(len(self.flats)):
            st = {"m": self.full_state("m", i).detach().cpu().clone(), "v": self.full_state("v", i).detach().cpu().clone()}
            if i >= 1:
                st["master"] = self.full_state("master", i).detach().cpu().clone()
            stores.append(st)
        sd["vlb"] = {"step_count": self.step_count, "stores": stores}
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        vlb = state_dict.pop("vlb", None)
        super().load_state_dict(state_dict)
        if vlb is None:
            raise KeyError("VlbAdamW.load_state_dict: no 'vlb' entry (moments / step count) - not a checkpoint of this optimiser")
        if len(vlb["stores"]) != len(self.flats):
            raise ValueError(f"checkpoint holds {len(vlb['stores'])} flat stores, this optimiser {len(self.flats)}")
        self.step_count = int(vlb["step_count"])
        for i, st in enumerate(vlb["stores"]):
            f = self.flats[i]
            if i == 0:
                self.load_full_state("master", f.master, 0)       # the module's load_state_dict restored the masters
            else:
                self.load_full_state("master", st["master"], i)
                self.compute_from_master(i)
            self.load_full_state("m", st["m"], i)
            self.load_full_state("v", st["v"], i)
        for fn in self.post_step:
            fn()                      # derived layouts (LoRA A^T / B pads, W^T copies) from the restored weights

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:               # Lightning's automatic optimisation: training_step + zero_grad + backward
            with torch.enable_grad():
                loss = closure()
        group = self.param_groups[0]
        self.step_count += 1
        st = _stream()
        for sh in self.shardeds:
            if sh is not None:
                sh.finish_reduce()           # reduce-scatters started under backward + the rest
        bufs = [sh if sh is not None else f for f, sh in zip(self.flats, self.shardeds)]
        self.sumsq.zero_()
        b1, b2 = group["betas"]
        if self.max_norm > 0:
            for f, b in zip(self.flats, bufs):
                fn = lib.vlb_grad_sumsq_bf16 if getattr(f, "grad_bf16", False) else lib.vlb_grad_sumsq
                check(fn(b.grad.data_ptr(), b.numel, self.sumsq.data_ptr(), self.sumsq_ws.data_ptr(), st), "vlb_grad_sumsq")
            active = [sh for sh in self.shardeds if sh is not None]
            if active:
                active[0].all_reduce_scalar(self.sumsq)      # the clip norm covers every rank's slices
        for f, b in zip(self.flats, bufs):
            fn = lib.vlb_adamw_step_g16 if getattr(f, "grad_bf16", False) else lib.vlb_adamw_step
            check(fn(b.master.data_ptr(), b.compute.data_ptr(), b.grad.data_ptr(), b.m.data_ptr(), b.v.data_ptr(), b.numel,
                     float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), self.step_count,
                     self.sumsq.data_ptr(), self.max_norm, st), "vlb_adamw_step")
        for sh in self.shardeds:
            if sh is not None:
                sh.gather_compute()
        for fn in self.post_step:
            fn()
        return loss

    def grad_norm(self) -> float:
        """Global gradient norm of the last step (host sync; for logging only)."""
        return float(self.sumsq.sqrt().item())
