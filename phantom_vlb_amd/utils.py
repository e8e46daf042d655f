"""Mirror of the reference's ``src/utils.py`` exports (HRFConvolveLayer, RidgeRegressionLayer,
get_hrf_weight, LogValAccuracyCallback; src/utils.py:14-110, src/__init__.py:3-13).

On the training path these layers do not run separately - the fused vlb_head_fwd / vlb_head_bwd
kernels do (phantom_vlb_amd/head.py).  The classes below keep the reference's names and call
signatures for code that uses the layers directly; they still execute on libvlb.
"""
from __future__ import annotations

import torch

from . import ops  # noqa: F401
from ._lib import check, lib


try:                                   # a real Callback when Lightning is installed: its Trainer calls every hook by name
    from lightning.pytorch.callbacks import Callback as _Callback
except Exception:
    class _Callback:
        """Hook-name compatible stand-in for ``lightning.pytorch.callbacks.Callback`` (the built-in Trainer calls the hooks
        that exist, by name)."""


def get_hrf_weight(time_diff: float) -> float:
    """src/utils.py:14-37 - preprocessing-time helper (Glover HRF).  nilearn when importable, else the
    restatement of its algorithm in phantom_vlb_amd/episodes.py (unpinned)."""
    from .episodes import get_hrf_weight as _impl
    return _impl(time_diff)


def _stream():
    return torch.cuda.current_stream().cuda_stream


class HRFConvolveLayer:
    """``torch.einsum('bse,bs->be', embeddings, hrf_weights)`` (src/utils.py:44-56) as one HBM-bound pass over the embeddings
    (vlb_hrf_pool: fp32 accumulation in a fixed order, tokens with zero weight skipped).  Returns the embeddings' dtype,
    like the reference's einsum; bf16 and fp32 embeddings are read as they are, anything else goes through fp32."""

    def __call__(self, embeddings, hrf_weights):
        return self.forward(embeddings, hrf_weights)

    def forward(self, embeddings, hrf_weights):
        if not embeddings.is_cuda:
            raise RuntimeError("phantom_vlb_amd layers need GPU tensors (no CPU fallback)")
        B, S, E = embeddings.shape
        assert tuple(hrf_weights.shape) == (B, S) and E % 8 == 0
        dt = embeddings.dtype
        x = embeddings.contiguous() if dt in (torch.bfloat16, torch.float32) else embeddings.float().contiguous()
        w = hrf_weights.to(embeddings.device, torch.float32).contiguous()
        out = torch.empty(B, E, dtype=x.dtype, device=x.device)
        ws = torch.empty(lib.vlb_hrf_pool_ws_floats(B, E), dtype=torch.float32, device=x.device)
        check(lib.vlb_hrf_pool(x.data_ptr(), int(x.dtype == torch.float32), w.data_ptr(), out.data_ptr(), ws.data_ptr(), B, S, E,
                               _stream()), "vlb_hrf_pool")
        return out.to(dt)


class _LinearParams:
    """``ridge_layer.linear``: the weight / bias holder whose attribute names the reference's state dict uses."""

    def __init__(self, weight, bias):
        self.weight, self.bias = weight, bias
        self.out_features, self.in_features = weight.shape


class RidgeRegressionLayer:
    """``nn.Linear(input_dim, output_dim, bias=True, **kwargs)`` + ``l2_lambda * ||W||_F^2`` (src/utils.py:59-73) on libvlb
    (vlb_ridge_fwd: one pass over W gives the prediction and sum(W^2)).  ``dtype`` defaults to bf16, what the reference's
    module passes (litmodule :218-223); the kernels read bf16 operands and accumulate in fp32; the output takes x's dtype and the
    penalty is an fp32 scalar (``torch.norm`` runs in fp32 under the reference's bf16-mixed autocast)."""

    def __init__(self, input_dim, output_dim, l2_lambda=0.01, device="cuda", dtype=torch.bfloat16, **kwargs):
        import math
        bound = 1.0 / math.sqrt(input_dim)              # nn.Linear default init
        self.l2_lambda = l2_lambda
        self.linear = _LinearParams(((torch.rand(output_dim, input_dim, device=device) * 2 - 1) * bound).to(dtype),
                                    ((torch.rand(output_dim, device=device) * 2 - 1) * bound).to(dtype))

    def __call__(self, x, add_regularization=True):
        return self.forward(x, add_regularization)

    def forward(self, x, add_regularization=True):
        if not x.is_cuda:
            raise RuntimeError("phantom_vlb_amd layers need GPU tensors (no CPU fallback)")
        W, b = self.linear.weight, self.linear.bias
        V, E = W.shape
        lead = x.shape[:-1]
        x2 = x.reshape(-1, E).to(torch.bfloat16).contiguous()
        B = x2.shape[0]
        pred = torch.empty(B, V, dtype=torch.float32, device=x.device)
        l2 = torch.empty(1, dtype=torch.float32, device=x.device)
        ws = torch.empty(lib.vlb_ridge_ws_floats(V), dtype=torch.float32, device=x.device)
        check(lib.vlb_ridge_fwd(x2.data_ptr(), W.to(torch.bfloat16).contiguous().data_ptr(), b.to(torch.bfloat16).contiguous().data_ptr(),
                                pred.data_ptr(), l2.data_ptr(), ws.data_ptr(), B, E, V, float(self.l2_lambda), _stream()), "vlb_ridge_fwd")
        out = pred.to(x.dtype).reshape(*lead, V)
        return (out, l2[0]) if add_regularization else out


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

    def all_reduce_sums(self, pl_module=None):
        """Data parallel validation (rank-strided batches): sum the running sums over ranks, once per epoch.  EVERY rank
        issues the same two collectives - a rank that drew no batch (fewer validation batches than ranks,
        ``limit_val_batches`` < world) contributes zeros instead of skipping them."""
        import torch.distributed as dist
        if self.s is None:
            if pl_module is None:
                raise RuntimeError("all_reduce_sums() on a rank without validation batches needs the module "
                                   "(device and num_target of the zero contribution)")
            self.s = torch.zeros(5, pl_module.config.num_target, dtype=torch.float64, device=pl_module.device)
        n = torch.tensor([float(self.n)], dtype=torch.float64, device=self.s.device)
        dist.all_reduce(self.s)
        dist.all_reduce(n)
        self.n = int(n.item())
        if self.n == 0:
            self.s = None                     # nobody validated anything: nothing to report

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
