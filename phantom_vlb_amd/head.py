"""Brain head on libvlb: LN1 -> HRF-weighted pool -> LN2 -> dropout -> ridge linear -> MSE + lambda*||W||^2.

Mirrors the reference's head modules (``layer_norm1``, ``hrf_layer``, ``layer_norm2``, ``dropout``,
``ridge_layer.linear``; src/litmodule/videollama2_vlb_litmodule.py:210-226,245-254 and
src/utils.py:40-73) but runs as the fused vlb_head_fwd / vlb_head_bwd kernels.
Trainable parameters keep an fp32 master (what AdamW updates) and the bf16 copy the kernels read -
the reference stores them in bf16 (``dtype=self.config.dtype``), i.e. the kernels see the same values.
"""
from __future__ import annotations

import math

import torch

from ._lib import check, lib

BF16 = torch.bfloat16
HEAD_PARAMS = ("layer_norm1.weight", "layer_norm1.bias", "layer_norm2.weight", "layer_norm2.bias",
               "ridge_layer.linear.weight", "ridge_layer.linear.bias")


def _stream():
    return torch.cuda.current_stream().cuda_stream


class BrainHead:
    def __init__(self, dim: int, num_target: int, l2_lambda: float, eps: float, device, sd: dict | None = None,
                 seed: int = 1234):
        self.E, self.V, self.l2_lambda, self.eps, self.dev = dim, num_target, float(l2_lambda), float(eps), device
        self.master: dict[str, torch.Tensor] = {}
        if sd is None:
            gen = torch.Generator(device="cpu").manual_seed(seed)
            bound = 1.0 / math.sqrt(dim)        # nn.Linear default init (kaiming_uniform a=sqrt 5)
            sd = {
                "layer_norm1.weight": torch.ones(dim), "layer_norm1.bias": torch.zeros(dim),
                "layer_norm2.weight": torch.ones(dim), "layer_norm2.bias": torch.zeros(dim),
                "ridge_layer.linear.weight": (torch.rand(num_target, dim, generator=gen) * 2 - 1) * bound,
                "ridge_layer.linear.bias": (torch.rand(num_target, generator=gen) * 2 - 1) * bound,
            }
            # the reference creates these modules in bf16 (dtype=self.config.dtype): start from bf16-representable values
            sd = {n: t.to(BF16).float() for n, t in sd.items()}
        for n in HEAD_PARAMS:
            # handed-in tensors are taken as they are (bf16 checkpoints widen exactly; an fp32 checkpoint of the
            # masters resumes at full precision)
            self.master[n] = sd[n].detach().to(device=device, dtype=torch.float32).contiguous()
        self.compute = {n: t.to(BF16) for n, t in self.master.items()}
        self.grads = {n: torch.zeros_like(t) for n, t in self.master.items()}
        self._cap = None

    def _buffers(self, B, S):
        key = (B, S)
        if self._cap != key:
            d, E, V = self.dev, self.E, self.V
            f32 = torch.float32
            self.ws = torch.empty(lib.vlb_head_ws_floats(B, S, E, V), dtype=f32, device=d)
            self.stats = torch.zeros(B, S, 2, dtype=f32, device=d)
            self.pooled_raw = torch.empty(B, E, dtype=f32, device=d)
            self.sumw = torch.empty(B, dtype=f32, device=d)
            self.zhat = torch.empty(B, E, dtype=f32, device=d)
            self.ln2_rstd = torch.empty(B, dtype=f32, device=d)
            self.z = torch.empty(B, E, dtype=BF16, device=d)
            self.pred = torch.empty(B, V, dtype=f32, device=d)
            self.loss_terms = torch.empty(3, dtype=f32, device=d)
            self.dz = torch.empty(B, E, dtype=f32, device=d)
            self.dpooled = torch.empty(B, E, dtype=f32, device=d)
            self._cap = key

    def forward(self, hidden, wmask, y, keep_scale=None, layout=None):
        """hidden bf16 [B*S,E] (or [B,S,E]; packed rows with a packed ``layout``), wmask f32 [B,S] (always
        dense), y f32 [B,V] -> (pred f32 [B,V], loss_terms f32[3])."""
        B, S = wmask.shape
        packed = layout is not None and layout.packed
        rows = layout.rows if packed else B * S
        if hidden.numel() != rows * self.E or (packed and (layout.B, layout.S) != (B, S)):
            raise ValueError(f"head: hidden has {hidden.numel() // self.E} rows, layout expects {rows}")
        cu = layout.cu if packed else None
        self._buffers(B, S)
        self._saved = (hidden, wmask, y, keep_scale, cu, rows)
        c = self.compute
        check(lib.vlb_head_fwd(hidden.data_ptr(), wmask.data_ptr(), c["layer_norm1.weight"].data_ptr(),
                               c["layer_norm1.bias"].data_ptr(), c["layer_norm2.weight"].data_ptr(),
                               c["layer_norm2.bias"].data_ptr(), c["ridge_layer.linear.weight"].data_ptr(),
                               c["ridge_layer.linear.bias"].data_ptr(), y.data_ptr(),
                               None if keep_scale is None else keep_scale.data_ptr(), self.ws.data_ptr(),
                               self.stats.data_ptr(), self.pooled_raw.data_ptr(), self.sumw.data_ptr(),
                               self.zhat.data_ptr(), self.ln2_rstd.data_ptr(), self.z.data_ptr(), self.pred.data_ptr(),
                               self.loss_terms.data_ptr(), B, S, self.E, self.V, self.eps, self.l2_lambda,
                               None if cu is None else cu.data_ptr(), _stream()),
              "vlb_head_fwd")
        return self.pred, self.loss_terms

    def backward(self, need_dhidden: bool, loss_scale: float = 1.0, l2_scale: float = 1.0):
        """Fills self.grads (fp32, overwritten) and returns d loss / d hidden (bf16) or None."""
        hidden, wmask, y, keep_scale, cu, rows = self._saved
        B, S = wmask.shape
        c, gr = self.compute, self.grads
        dh = torch.empty(rows, self.E, dtype=BF16, device=self.dev) if need_dhidden else None
        check(lib.vlb_head_bwd(hidden.data_ptr(), wmask.data_ptr(), c["layer_norm1.weight"].data_ptr(),
                               c["layer_norm2.weight"].data_ptr(), c["ridge_layer.linear.weight"].data_ptr(),
                               y.data_ptr(), None if keep_scale is None else keep_scale.data_ptr(),
                               self.stats.data_ptr(), self.pooled_raw.data_ptr(), self.sumw.data_ptr(),
                               self.zhat.data_ptr(), self.ln2_rstd.data_ptr(), self.z.data_ptr(), self.pred.data_ptr(),
                               gr["ridge_layer.linear.weight"].data_ptr(), gr["ridge_layer.linear.bias"].data_ptr(),
                               gr["layer_norm2.weight"].data_ptr(), gr["layer_norm2.bias"].data_ptr(),
                               gr["layer_norm1.weight"].data_ptr(), gr["layer_norm1.bias"].data_ptr(),
                               self.ws.data_ptr(), self.dz.data_ptr(), self.dpooled.data_ptr(),
                               None if dh is None else dh.data_ptr(), B, S, self.E, self.V, self.eps, self.l2_lambda,
                               float(loss_scale), float(l2_scale), None if cu is None else cu.data_ptr(), rows,
                               _stream()), "vlb_head_bwd")
        return dh
