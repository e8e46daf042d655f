"""Mirror of the reference's ``src/utils.py`` exports (HRFConvolveLayer, RidgeRegressionLayer,
get_hrf_weight, LogValAccuracyCallback; src/utils.py:14-110, src/__init__.py:3-13).

On the training path these layers do not run separately - the fused vlb_head_fwd / vlb_head_bwd
kernels do (phantom_vlb_amd/head.py).  The classes below keep the reference's names and call
signatures for code that uses the layers directly; they still execute on libvlb.
"""
from __future__ import annotations

import torch

from . import ops


class _Callback:
    """Hook-name compatible stand-in for ``lightning.pytorch.callbacks.Callback`` (the built-in Trainer calls hooks by
    name; the real base class is never needed - see litmodule._Base)."""


def get_hrf_weight(time_diff: float) -> float:
    """src/utils.py:14-37 - preprocessing-time helper (Glover HRF).  nilearn when importable, else the
    restatement of its algorithm in phantom_vlb_amd/episodes.py (unpinned)."""
    from .episodes import get_hrf_weight as _impl
    return _impl(time_diff)


class HRFConvolveLayer:
    """einsum('bse,bs->be', embeddings, hrf_weights) (src/utils.py:44-56) as an M=1 GEMM per clip."""

    def __call__(self, embeddings, hrf_weights):
        return self.forward(embeddings, hrf_weights)

    def forward(self, embeddings, hrf_weights):
        B, S, E = embeddings.shape
        emb_t = embeddings.to(torch.bfloat16).transpose(1, 2).contiguous()      # [B,E,S]: rows = output columns
        w = torch.zeros(B, 8, (S + 7) // 8 * 8, dtype=torch.bfloat16, device=embeddings.device)
        w[:, 0, :S] = hrf_weights.to(torch.bfloat16)
        out = torch.empty(B, E, dtype=torch.bfloat16, device=embeddings.device)
        for b in range(B):
            a = emb_t[b]
            if S % 8:
                pad = torch.zeros(E, w.shape[2], dtype=torch.bfloat16, device=a.device)
                pad[:, :S] = a
                a = pad
            out[b] = ops.gemm(w[b], a)[0]
        return out


class RidgeRegressionLayer:
    """Linear(input_dim -> output_dim, bias) + l2_lambda * ||W||_F^2 (src/utils.py:59-73)."""

    def __init__(self, input_dim, output_dim, l2_lambda=0.01, device="cuda", dtype=torch.bfloat16, **kwargs):
        import math
        bound = 1.0 / math.sqrt(input_dim)
        self.l2_lambda = l2_lambda
        self.weight = ((torch.rand(output_dim, input_dim, device=device) * 2 - 1) * bound).to(dtype)
        self.bias = ((torch.rand(output_dim, device=device) * 2 - 1) * bound).to(dtype)
        self.linear = self

    def __call__(self, x, add_regularization=True):
        return self.forward(x, add_regularization)

    def forward(self, x, add_regularization=True):
        B = x.shape[0]
        xp = torch.zeros((B + 7) // 8 * 8, x.shape[1], dtype=torch.bfloat16, device=x.device)
        xp[:B] = x
        out = ops.gemm(xp, self.weight, bias=self.bias)[:B]
        if add_regularization:
            return out, self.l2_lambda * self.weight.float().pow(2).sum()
        return out


class LogValAccuracyCallback(_Callback):
    """Per-target Pearson r over the validation set (src/utils.py:85-110), streamed: five running sums
    per target on the device instead of concatenating every prediction, and ONE tensor of per-target
    correlations (kept available as `correlations`) copied to the host once before the per-ROI log() calls."""

    def on_validation_epoch_start(self, trainer, pl_module):
        self.n = 0
        self.s = None

    def on_validation_batch_end(self, trainer, pl_module, outputs, batch, batch_idx, dataloader_idx=0):
        y = torch.nan_to_num(outputs["brain_vals"].double())
        p = torch.nan_to_num(outputs["brain_preds"].double())
        cur = torch.stack([p.sum(0), y.sum(0), (p * p).sum(0), (y * y).sum(0), (p * y).sum(0)])
        self.s = cur if self.s is None else self.s + cur
        self.n += p.shape[0]

    def all_reduce_sums(self):
        """Data parallel validation (rank-strided batches): sum the running sums over ranks, once per epoch."""
        import torch.distributed as dist
        if self.s is None:
            return
        n = torch.tensor([float(self.n)], dtype=torch.float64, device=self.s.device)
        dist.all_reduce(self.s)
        dist.all_reduce(n)
        self.n = int(n.item())

    def on_validation_epoch_end(self, trainer, pl_module):
        if self.s is None:
            return
        n = float(self.n)
        sp, sy, spp, syy, spy = self.s
        cov = spy - sp * sy / n
        var = (spp - sp * sp / n) * (syy - sy * sy / n)
        self.correlations = (cov / var.clamp_min(1e-30).sqrt()).float()
        # one scalar per target, like the reference (:105-110) - from ONE device->host copy instead of num_target syncs;
        # max_roi_logs caps the count for whole-cortex heads (65,536 log() calls per validation epoch otherwise)
        host = self.correlations.cpu().tolist()
        limit = getattr(self, "max_roi_logs", None)
        for i in range(pl_module.config.num_target if limit is None else min(pl_module.config.num_target, limit)):
            pl_module.log(f"val_corr_ROI_{i:0{6}}", host[i])
        pl_module.log("val_corr_avg", self.correlations.mean())
