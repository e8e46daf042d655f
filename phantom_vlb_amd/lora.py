"""LoRA fine-tuning of the decoder on libvlb: explicit forward-with-save and backward.

What the reference gets from ``peft.get_peft_model(model, LoraConfig(task_type="FEATURE_EXTRACTION",
r, lora_alpha, lora_dropout, target_modules=find_all_linear_names(model)))``
(src/litmodule/videollama2_vlb_litmodule.py:36-55,113-120): every decoder linear
``{q,k,v,o,gate,up,down}_proj`` becomes ``y = x W^T + (alpha/r) * B(A(dropout_p(x)))`` with W frozen,
A kaiming-uniform(a=sqrt 5), B zero.  (peft matches target names by suffix, so upstream the CLIP
``q_proj/k_proj/v_proj`` could be wrapped too; the tower is frozen and BASELINE.json says "LoRA on
attn/MLP", so only the 7 decoder linears are adapted - SURVEY.md 7.2.)

MI355X layout: projections that share an input share one skinny launch (at r = 16: q,k,v -> R=48; gate,up ->
R=32; larger ranks are stacks of 16-rank blocks, 48 ranks per launch); the adapter's up-projection rides in the base
GEMM as a second operand pair (t | B), so
adapted and base outputs are accumulated in the same MFMA accumulators; dgrad uses the transposed
frozen weights laid down once at load time.  Masters are fp32 ``A [r,in]`` and ``B^T [r,out]``.
"""
from __future__ import annotations

import ctypes
import math

import torch

from . import ops
from ._lib import check, lib
from .geometry import Geometry

BF16 = torch.bfloat16
PAD = 64  # adapter rank columns padded to one GEMM K-tile
FUSE_SWIGLU_BWD = True      # SwiGLU backward in the epilogue of the down projection's dgrad GEMM (A/B switch)
MERGE_DB_U = True           # one dB^T / u sweep for the projections sharing a dy (q|k|v, gate|up) instead of one each (A/B switch)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _seeds(vals):
    return (ctypes.c_uint32 * len(vals))(*[v & 0xFFFFFFFF for v in vals])


def lora_down(x, A, R, scale, p, seeds, out):
    M, K = x.shape
    check(lib.vlb_lora_down(x.data_ptr(), x.stride(0), A.data_ptr(), out.data_ptr(), out.stride(0), M, K, R, scale, p,
                            _seeds(seeds) if p > 0 else None, _stream()), "vlb_lora_down")
    return out


def lora_dx_masked(u, At, dx, R, p, seeds):
    M, K = dx.shape
    check(lib.vlb_lora_dx_masked(u.data_ptr(), u.stride(0), At.data_ptr(), At.stride(0), dx.data_ptr(), dx.stride(0), M, K, R, p,
                                 _seeds(seeds), _stream()), "vlb_lora_dx_masked")


def wgrad_skinny_u(G, X, dW, ws, Bt, u_scale, u_out, u_ws, alpha=1.0, beta=0.0):
    """dW [16,K] fp32 = alpha G[:, :16]^T X + beta dW  and  u_out[:, :16] = u_scale X Bt^T in one pass over X."""
    M, K = X.shape
    check(lib.vlb_wgrad_skinny_u(G.data_ptr(), G.stride(0), X.data_ptr(), X.stride(0), dW.data_ptr(), ws.data_ptr(), M, K,
                                 alpha, beta, Bt.data_ptr(), u_scale, u_out.data_ptr(), u_out.stride(0), u_ws.data_ptr(),
                                 _stream()), "vlb_wgrad_skinny_u")


def wgrad_skinny_u_multi(G, X, cols, dWs, Bts, ws, u_scale, u_out, u_ws, alpha=1.0, beta=0.0):
    """The dB^T / u pass of ``wgrad_skinny_u`` for the projections sharing X = dy [M, sum(cols)] in one sweep: projection j
    reads t = G[:, 16j:16j+16], writes dWs[j] [16, cols[j]] fp32 and u_out[:, 16j:16j+16]."""
    M = X.shape[0]
    n = len(cols)
    check(lib.vlb_wgrad_skinny_u_multi(G.data_ptr(), G.stride(0), X.data_ptr(), X.stride(0), M, n, (ctypes.c_int * n)(*cols),
                                       (ctypes.c_void_p * n)(*[t.data_ptr() for t in dWs]), (ctypes.c_void_p * n)(*[t.data_ptr() for t in Bts]),
                                       ws.data_ptr(), alpha, beta, u_scale, u_out.data_ptr(), u_out.stride(0), u_ws.data_ptr(), _stream()),
          "vlb_wgrad_skinny_u_multi")


def wgrad_skinny(G, X, dW, ws, N, alpha=1.0, beta=0.0, p=0.0, seeds=None):
    """dW [N,K] fp32 (contiguous) = alpha/(1-p) * G[:, :N]^T keep(X) + beta*dW; N in {16,32,48}."""
    M, K = X.shape
    check(lib.vlb_wgrad_skinny(G.data_ptr(), G.stride(0), X.data_ptr(), X.stride(0), dW.data_ptr(), ws.data_ptr(), M, N, K,
                               alpha, beta, p, _seeds(seeds) if p > 0 else None, _stream()), "vlb_wgrad_skinny")


# (group name, [(target, out rows attr)], input) - projections in one group share their input
GROUPS = (
    ("qkv", ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj")),
    ("o", ("self_attn.o_proj",)),
    ("gu", ("mlp.gate_proj", "mlp.up_proj")),
    ("down", ("mlp.down_proj",)),
)


class LoraState:
    def __init__(self, g: Geometry, weights, r: int, alpha: int, dropout: float, device, seed: int = 1234,
                 sd: dict | None = None, target_modules=None):
        """``r``: any rank 1..64 (the reference's field is a free int, litmodule :141; its YAML sets 16, upstream
        VideoLLaMA2 recipes 64).  The adapter kernels work on 16-wide MFMA tiles, so a projection's adapter is
        ``c = ceil(r/16)`` stacked 16-rank blocks (same dropout mask: same seed) and the last block is zero-padded:
        padded rows of A and B^T are zero, receive exactly zero gradients (t = x.0, u = dy.0) and stay zero under
        AdamW; ``state_dict`` / ``load_state_dict`` expose the true [r, in] / [out, r] shapes.
        ``target_modules``: leaf names from ``find_all_linear_names`` - must be the seven decoder linears."""
        if not isinstance(r, int) or not 1 <= r <= 64:
            raise NotImplementedError(f"lora_r={r}: ranks 1..64 are built (up to four 16-wide MFMA tiles per adapted projection); "
                                      "the reference's setting is 16 (config/experiment/VLB_vllama2_friends_lora.yaml:27)")
        want = sorted(t.split(".")[-1] for _, ts in GROUPS for t in ts)
        if target_modules is not None and sorted(target_modules) != want:
            raise NotImplementedError(f"LoRA target_modules {sorted(target_modules)}: the adapted decoder is built for "
                                      f"exactly {want}")
        self.g, self.w, self.r, self.dev = g, weights, r, device
        # gate/up weight rows interleaved for the SwiGLU epilogue (Weights(gate_up_interleaved=True)): the adapter's padded
        # B rows follow the same interleave and the forward is ONE GEMM that writes silu(gate)*up and keeps [gate | up]
        self.gu_il = "wgu_il" in weights.layers[0]
        self.c = c = (r + 15) // 16          # 16-rank blocks per projection
        self.rp = rp = 16 * c                # padded rank (rows of A / B^T per projection)
        self.scale = alpha / r
        self.p = float(dropout)
        self.step = 0
        self.rank = 0                  # data-parallel rank, mixed into the dropout seeds (ranks draw distinct masks)
        self.grad_hook = None          # callable(layer index) fired by backward when a layer's gradients are final
        self.base_seed = seed
        qd, kd = g.heads * g.head_dim, g.kv_heads * g.head_dim
        self.out_dims = {"self_attn.q_proj": qd, "self_attn.k_proj": kd, "self_attn.v_proj": kd, "self_attn.o_proj": g.dim,
                         "mlp.gate_proj": g.ff, "mlp.up_proj": g.ff, "mlp.down_proj": g.dim}
        self.in_dims = {"self_attn.q_proj": g.dim, "self_attn.k_proj": g.dim, "self_attn.v_proj": g.dim,
                        "self_attn.o_proj": qd, "mlp.gate_proj": g.dim, "mlp.up_proj": g.dim, "mlp.down_proj": g.ff}
        gen = torch.Generator(device="cpu").manual_seed(seed + 17)
        self.master: dict[str, torch.Tensor] = {}
        self.grads: dict[str, torch.Tensor] = {}
        self.layers = []
        for i in range(g.layers):
            lay = {}
            for gname, targets in GROUPS:
                R = rp * len(targets)
                Rpad = (R + PAD - 1) // PAD * PAD                                  # K2 of the fused GEMM: whole 64-wide K-tiles
                kin = self.in_dims[targets[0]]
                A = torch.zeros(R, kin, dtype=BF16, device=device)                 # stacked adapters (compute copy)
                nout = sum(self.out_dims[t] for t in targets)
                lay[gname] = dict(A=A, At=torch.zeros(kin, Rpad, dtype=BF16, device=device),
                                  Bpad=torch.zeros(nout, Rpad, dtype=BF16, device=device), R=R, Rpad=Rpad, targets=targets)
                for j, t in enumerate(targets):
                    pre = f"model.layers.{i}.{t}"
                    a0, b0 = torch.zeros(rp, kin), torch.zeros(rp, self.out_dims[t])
                    if sd is not None and f"{pre}.lora_A.weight" in sd:
                        a_in, b_in = sd[f"{pre}.lora_A.weight"].float(), sd[f"{pre}.lora_B.weight"].float()
                        if tuple(a_in.shape) != (r, kin) or tuple(b_in.shape) != (self.out_dims[t], r):
                            raise ValueError(f"{pre}: adapter shapes {tuple(a_in.shape)}/{tuple(b_in.shape)} do not match "
                                             f"lora_r={r} (peft layout: lora_A [r,in], lora_B [out,r])")
                        a0[:r], b0[:r] = a_in, b_in.t()
                    else:
                        bound = 1.0 / math.sqrt(kin)           # kaiming_uniform(a=sqrt(5))
                        a0[:r] = (torch.rand(rp, kin, generator=gen)[:r] * 2 - 1) * bound
                    self.master[f"{pre}.lora_A.weight"] = a0.to(device).contiguous()
                    self.master[f"{pre}.lora_B.weight"] = b0.to(device).contiguous()    # stored as B^T [r,out]
            self.layers.append(lay)
        # gradients: A-grads of a group are row-slices of one stacked [R,in] buffer (one wgrad launch per group)
        self.grads = {}
        self.grad_A = []
        for i in range(g.layers):
            ga = {}
            for gname, targets in GROUPS:
                buf = torch.zeros(rp * len(targets), self.in_dims[targets[0]], dtype=torch.float32, device=device)
                ga[gname] = buf
                for j, t in enumerate(targets):
                    self.grads[f"model.layers.{i}.{t}.lora_A.weight"] = buf[rp * j:rp * j + rp]
                    self.grads[f"model.layers.{i}.{t}.lora_B.weight"] = torch.zeros(rp, self.out_dims[t], dtype=torch.float32, device=device)
            self.grad_A.append(ga)
        # bf16 compute copies the optimiser refreshes in place: A rows inside the stacked matrix, B^T separate
        self.bt = {n: torch.zeros(t.shape, dtype=BF16, device=device) for n, t in self.master.items() if "lora_B" in n}
        self.refresh(from_master=True)
        self._ws = None

    # ------------------------------------------------------------------ parameter plumbing
    def named_masters(self):
        return list(self.master.items())

    def compute_copies(self):
        out = {}
        for i, lay in enumerate(self.layers):
            for gname, targets in GROUPS:
                for j, t in enumerate(targets):
                    pre = f"model.layers.{i}.{t}"
                    out[f"{pre}.lora_A.weight"] = lay[gname]["A"][self.rp * j:self.rp * j + self.rp]
                    out[f"{pre}.lora_B.weight"] = self.bt[f"{pre}.lora_B.weight"]
        return out

    def _scatter_jobs(self):
        """Device table for vlb_transpose16_scatter: every A_j -> At band and B_j^T -> Bpad band, cut in
        256-row jobs.  Built lazily and rebuilt when the compute copies are re-pointed (FlatTrainables)."""
        import struct
        cc = self.compute_copies()
        key = tuple(t.data_ptr() for t in cc.values())
        if getattr(self, "_jobs_key", None) == key:
            return self._jobs
        rec = []
        for i, lay in enumerate(self.layers):
            for gname, targets in GROUPS:
                blk = lay[gname]
                row = 0
                for j, t in enumerate(targets):
                    pre = f"model.layers.{i}.{t}"
                    a, bt = cc[f"{pre}.lora_A.weight"], cc[f"{pre}.lora_B.weight"]
                    for b in range(self.c):            # one 16-row block at a time (the kernel transposes 16 x 256 pieces)
                        col = self.rp * j + 16 * b
                        il = int(self.gu_il and gname == "gu")       # B rows of gate / up land interleaved in 16-row blocks
                        brow = 16 * j if il else row
                        for src, dst, n, mode in ((a[16 * b:16 * b + 16], blk["At"][:, col:], a.shape[1], 0),
                                                  (bt[16 * b:16 * b + 16], blk["Bpad"][brow:, col:], bt.shape[1], il)):
                            assert src.is_contiguous() and src.shape[0] == 16 and dst.stride(0) == blk["Rpad"]
                            for n0 in range(0, n, 256):
                                rec.append(struct.pack("<QQiiii", src.data_ptr(), dst.data_ptr(), n, n0, blk["Rpad"], mode))
                    row += self.out_dims[t]
        buf = torch.frombuffer(bytearray(b"".join(rec)), dtype=torch.uint8).to(self.dev)
        self._jobs, self._jobs_key, self._n_jobs = buf, key, len(rec)
        return buf

    def refresh(self, from_master=False):
        """Rebuild the derived layouts (A^T padded, block-diagonal padded B) from the bf16 copies: one
        launch over all layers (the per-tensor torch copies this replaces were ~2 % of a LoRA step)."""
        if from_master:
            cc = self.compute_copies()
            for n, t in cc.items():
                t.copy_(self.master[n])
        jobs = self._scatter_jobs()
        check(lib.vlb_transpose16_scatter(jobs.data_ptr(), self._n_jobs, _stream()), "vlb_transpose16_scatter")

    def state_dict(self):
        """peft layout: lora_A [r,in], lora_B [out,r] (rank padding removed)."""
        r = self.r
        return {n: (t[:r].t().contiguous() if "lora_B" in n else t[:r].clone()) for n, t in self.master.items()}

    def load_state_dict(self, sd: dict):
        """peft-layout tensors -> fp32 masters (padded rows stay zero), then bf16 copies + derived layouts."""
        r = self.r
        for n, t in self.master.items():
            src = sd[n].to(self.dev, torch.float32)
            t[:r].copy_(src.t() if "lora_B" in n else src)
        self.refresh(from_master=True)

    def _seed(self, layer, target_idx):
        x = (self.base_seed * 0x9E3779B1 + self.step * 0x85EBCA6B + layer * 0xC2B2AE35 + target_idx * 0x27D4EB2F
             + self.rank * 0x7F4A7C15) & 0xFFFFFFFF
        return x or 1

    def _workspace(self, M):
        if self._ws is None or self._ws["cap"] < M:          # grow-only: packed batches change M every step
            g, d = self.g, self.dev
            kmax = max(g.ff, g.dim, g.heads * g.head_dim)
            kgrp = max(2 * g.ff, (g.heads + 2 * g.kv_heads) * g.head_dim)      # widest dy a group's projections share
            self._ws = dict(cap=M, wg=torch.empty(lib.vlb_wgrad_splits(M) * 48 * kmax, dtype=torch.float32, device=d),
                            uws=torch.empty(lib.vlb_wgrad_u_ws_floats(M, kgrp), dtype=torch.float32, device=d),
                            u_full=torch.zeros(M, self._rpad_max(), dtype=BF16, device=d))
        self._ws["u"] = self._ws["u_full"][:M]
        return self._ws

    def _rpad_max(self):
        return max(blk["Rpad"] for blk in self.layers[0].values())

    def _group_seeds(self, seeds, n_targets):
        """One seed per 16-rank block: the c blocks of a projection share its dropout mask."""
        return None if seeds is None else [seeds[j] for j in range(n_targets) for _ in range(self.c)]

    # ------------------------------------------------------------------ forward with saved activations
    def _t_buffer(self, li, gi, M):
        """Persistent [M, 64] adapter activations per (layer, group): columns >= R stay zero from allocation
        (the fused GEMM's K2 = 64 operand), so no per-call fill; grow-only when a packed batch is larger."""
        if getattr(self, "_t_cap", 0) < M:
            self._t_cap = M
            self._t_bufs = torch.zeros(len(self.layers), len(GROUPS), M, self._rpad_max(), dtype=BF16, device=self.dev)
        return self._t_bufs[li, gi, :M, :self.layers[li][GROUPS[gi][0]]["Rpad"]]

    def _lora_t(self, x, blk, seeds, slot=None, p=None):
        """t = s/(1-p) * keep(x) . A^T for every projection of the group (the fused GEMM's second A operand)."""
        t = self._t_buffer(*slot, x.shape[0]) if slot is not None else torch.zeros(x.shape[0], blk["Rpad"], dtype=BF16, device=self.dev)
        p = self.p if p is None else p
        gs = self._group_seeds(seeds, len(blk["targets"])) if p > 0 else None
        for r0 in range(0, blk["R"], 48):                       # the skinny kernel takes up to three 16-rank blocks per launch
            n = min(48, blk["R"] - r0)
            lora_down(x, blk["A"][r0:r0 + n], n, self.scale, p, None if gs is None else gs[r0 // 16:(r0 + n) // 16], t[:, r0:])
        return t

    def _adapted(self, x, W, blk, seeds, residual=None, slot=None, p=None):
        t = self._lora_t(x, blk, seeds, slot, p)
        return ops.gemm(x, W, residual=residual, a2=t, w2=blk["Bpad"]), t

    def _adapted_mlp_in(self, x, lw, blk, seeds, slot, p=None, save=True):
        """Adapted gate/up projection + SwiGLU: returns (silu(gate)*up, [gate | up] or None, t)."""
        if self.gu_il:
            t = self._lora_t(x, blk, seeds, slot, p)
            if save:
                hh, gu = ops.gemm_swiglu_save(x, lw["wgu_il"], a2=t, w2_il=blk["Bpad"])
                return hh, gu, t
            return ops.gemm(x, lw["wgu_il"], act=ops.ACT_SWIGLU_PAIR, a2=t, w2=blk["Bpad"]), None, t
        gu, t = self._adapted(x, lw["wgu"], blk, seeds, slot=slot, p=p)
        return ops.swiglu(gu), gu, t

    def forward(self, backbone, vision_f32, ids, layout=None, train=True):
        """Forward of the whole backbone with the adapters.  train=True: dropout on, decoder activations kept for
        backward.  train=False (validation; peft in eval mode): same adapters, no dropout, nothing kept.
        ``layout``: packed RowLayout (rows without the clips' padded tails) or None for dense [B,S]."""
        g = self.g
        B = vision_f32.shape[0]
        vid = backbone.video_tokens(vision_f32)          # frozen: no activations kept; may have been started one step ahead
        x, key_mask = backbone.splice(ids, vid, layout)
        return self.decoder_forward(backbone, x, key_mask, B, layout, train=train)

    def decoder_forward(self, backbone, x, key_mask, B, layout=None, train=True):
        """The adapted decoder on spliced embeddings x [rows, dim] (the part of forward() that keeps activations)."""
        g, w = self.g, self.w
        S = g.max_len
        pos = None if layout is None else layout.pos
        qd, kd = g.heads * g.head_dim, g.kv_heads * g.head_dim
        if not train:
            return self._decoder_eval(backbone, x, key_mask, B, layout)
        self.saved = []
        self.step += 1
        for li, lay in enumerate(self.layers):
            lw = backbone.layer_weights(li)
            sd = [self._seed(li, k) for k in range(7)]
            h1 = ops.rmsnorm(x, lw["in_norm"], g.rms_eps)
            qkv, t_qkv = self._adapted(h1, lw["wqkv"], lay["qkv"], sd[0:3], slot=(li, 0))
            ops.rope_(qkv, w.rope_cos, w.rope_sin, B, S, g.heads + g.kv_heads, g.head_dim, pos=pos)
            a, lse = ops.attention_fwd(qkv[:, :qd], qkv[:, qd:qd + kd], qkv[:, qd + kd:], B, S, g.heads, g.kv_heads,
                                       g.head_dim, True, g.head_dim ** -0.5, key_mask=key_mask, need_lse=True,
                                       layout=layout)
            x2, t_o = self._adapted(a, lw["wo"], lay["o"], sd[3:4], residual=x, slot=(li, 1))
            h2 = ops.rmsnorm(x2, lw["post_norm"], g.rms_eps)
            hh, gu, t_gu = self._adapted_mlp_in(h2, lw, lay["gu"], sd[4:6], (li, 2))
            x3, t_d = self._adapted(hh, lw["wdown"], lay["down"], sd[6:7], residual=x2, slot=(li, 3))
            self.saved.append(dict(x=x, h1=h1, qkv=qkv, a=a, lse=lse, x2=x2, h2=h2, gu=gu, hh=hh, t_qkv=t_qkv, t_o=t_o,
                                   t_gu=t_gu, t_d=t_d, seeds=sd))
            x = x3
        self.x_last, self.key_mask, self.B, self.layout = x, key_mask, B, layout
        return ops.rmsnorm(x, w.final_norm, g.rms_eps), key_mask

    def _decoder_eval(self, backbone, x, key_mask, B, layout):
        """Adapted decoder without dropout and without saved activations (validation / inference)."""
        g, w = self.g, self.w
        S = g.max_len
        pos = None if layout is None else layout.pos
        qd, kd = g.heads * g.head_dim, g.kv_heads * g.head_dim
        for li, lay in enumerate(self.layers):
            lw = backbone.layer_weights(li)
            h = ops.rmsnorm(x, lw["in_norm"], g.rms_eps)
            qkv, _ = self._adapted(h, lw["wqkv"], lay["qkv"], None, slot=(li, 0), p=0.0)
            ops.rope_(qkv, w.rope_cos, w.rope_sin, B, S, g.heads + g.kv_heads, g.head_dim, pos=pos)
            a = ops.attention_fwd(qkv[:, :qd], qkv[:, qd:qd + kd], qkv[:, qd + kd:], B, S, g.heads, g.kv_heads, g.head_dim,
                                  True, g.head_dim ** -0.5, key_mask=key_mask, layout=layout)
            x, _ = self._adapted(a, lw["wo"], lay["o"], None, residual=x, slot=(li, 1), p=0.0)
            h = ops.rmsnorm(x, lw["post_norm"], g.rms_eps)
            hh, _, _ = self._adapted_mlp_in(h, lw, lay["gu"], None, (li, 2), p=0.0, save=False)
            x, _ = self._adapted(hh, lw["wdown"], lay["down"], None, residual=x, slot=(li, 3), p=0.0)
        return ops.rmsnorm(x, w.final_norm, g.rms_eps), key_mask

    # ------------------------------------------------------------------ backward
    def _group_backward(self, li, gname, dy, x_in, t, seeds, W_t, need_dx, swiglu_gu=None):
        """dy: grad of the group's (concatenated) output; x_in: the group's input; returns d x_in - or, with
        ``swiglu_gu`` (the saved [gate | up] activations x_in = silu(gate)*up came from), d [gate | up]."""
        lay = self.layers[li][gname]
        ws = self._workspace(dy.shape[0])
        u = ws["u"]
        col = 0
        u = u[:, :lay["Rpad"]]
        rp = self.rp
        names = [f"model.layers.{li}.{tname}" for tname in lay["targets"]]
        cols = [self.out_dims[tname] for tname in lay["targets"]]
        multi = MERGE_DB_U and self.c == 1 and len(cols) > 1 and all(n % 256 == 0 for n in cols) and dy.shape[1] == sum(cols)
        if multi:           # q|k|v and gate|up: one sweep over the shared dy for every projection's dB^T and u
            wgrad_skinny_u_multi(t, dy, cols, [self.grads[f"{pre}.lora_B.weight"] for pre in names],
                                 [self.bt[f"{pre}.lora_B.weight"] for pre in names], ws["wg"], self.scale, u, ws["uws"])
        for j, tname in enumerate(() if multi else lay["targets"]):
            n = self.out_dims[tname]
            pre = f"model.layers.{li}.{tname}"
            dyj = dy[:, col:col + n]
            # one pass over dy_j per 16-rank block: dB^T[r,out] = sum_m t[m,r] dy[m,out] (t carries s and 1/(1-p))  and
            # u_j = s * dy_j . B_j for the dA / dx terms below
            for b in range(self.c):
                k = rp * j + 16 * b
                wgrad_skinny_u(t[:, k:k + 16], dyj, self.grads[f"{pre}.lora_B.weight"][16 * b:16 * b + 16], ws["wg"],
                               self.bt[f"{pre}.lora_B.weight"][16 * b:16 * b + 16], self.scale, u[:, k:k + 16], ws["uws"])
            col += n
        # dA[r,in] = sum_m u[m,r] keep_g(x[m,in])/(1-p) for every projection of the group in one pass over x (48 ranks per launch)
        gs = self._group_seeds(seeds, len(lay["targets"])) if self.p > 0 else None
        for r0 in range(0, lay["R"], 48):
            n = min(48, lay["R"] - r0)
            wgrad_skinny(u[:, r0:], x_in, self.grad_A[li][gname][r0:r0 + n], ws["wg"], n, p=self.p,
                         seeds=None if gs is None else gs[r0 // 16:(r0 + n) // 16])
        if not need_dx:
            return None
        if self.p == 0.0:
            dx = ops.gemm(dy, W_t, a2=u, w2=lay["At"])
        elif lay["R"] == 16 and ops.gemm_masked_pair_ok(dy.shape[0], W_t.shape[0], dy.shape[1]):
            # single-projection groups (o, down): the dropout mask is applied to the u.A accumulators inside the
            # dgrad GEMM - no read-modify-write pass over dx; for `down` the SwiGLU backward rides in its epilogue
            if swiglu_gu is not None and FUSE_SWIGLU_BWD:
                return ops.gemm_masked_pair_swiglu_bwd(dy, W_t, swiglu_gu, u, lay["At"], self.p, seeds[0])
            dx = ops.gemm_masked_pair(dy, W_t, u, lay["At"], self.p, seeds[0])
        else:
            dx = ops.gemm(dy, W_t)
            for r0 in range(0, lay["R"], 48):
                n = min(48, lay["R"] - r0)
                lora_dx_masked(u[:, r0:], lay["At"][:, r0:], dx, n, self.p, gs[r0 // 16:(r0 + n) // 16])
        return dx if swiglu_gu is None else ops.swiglu_bwd(swiglu_gu, dx)

    def backward(self, backbone, dhidden):
        """dhidden: d loss / d (post-final-norm hidden) bf16 [B*S, dim].  Fills self.grads."""
        g, w = self.g, self.w
        B, S = self.B, g.max_len
        qd, kd = g.heads * g.head_dim, g.kv_heads * g.head_dim
        layout = self.layout
        pos = None if layout is None else layout.pos
        M = self.x_last.shape[0]
        dx = ops.rmsnorm_bwd(self.x_last, w.final_norm, dhidden, g.rms_eps)
        delta = torch.empty(B, g.heads, S, dtype=torch.float32, device=self.dev)
        for li in range(g.layers - 1, -1, -1):
            lw, sv = backbone.layer_weights(li, transposed=True, direction=-1), self.saved[li]
            sd = sv["seeds"]
            d_gu = self._group_backward(li, "down", dx, sv["hh"], sv["t_d"], sd[6:7], lw["wdown_t"], True, swiglu_gu=sv["gu"])
            d_h2 = self._group_backward(li, "gu", d_gu, sv["h2"], sv["t_gu"], sd[4:6], lw["wgu_t"], True)
            dx2 = ops.rmsnorm_bwd(sv["x2"], lw["post_norm"], d_h2, g.rms_eps, dx_in=dx)
            d_a = self._group_backward(li, "o", dx2, sv["a"], sv["t_o"], sd[3:4], lw["wo_t"], True)
            qkv = sv["qkv"]
            dqkv = ops.attention_bwd(qkv, qd, kd, sv["a"], d_a, sv["lse"], self.key_mask, B, S, g.heads, g.kv_heads,
                                     g.head_dim, True, g.head_dim ** -0.5, layout=layout, delta=delta)
            ops.rope_(dqkv, w.rope_cos, w.rope_sin, B, S, g.heads + g.kv_heads, g.head_dim, sign=-1, pos=pos)
            need_dx = li > 0                # embeddings / connector are frozen: nothing upstream of layer 0 trains
            d_h1 = self._group_backward(li, "qkv", dqkv, sv["h1"], sv["t_qkv"], sd[0:3], lw["wqkv_t"], need_dx)
            if need_dx:
                dx = ops.rmsnorm_bwd(sv["x"], lw["in_norm"], d_h1, g.rms_eps, dx_in=dx2)
            if self.grad_hook is not None:
                self.grad_hook(li)            # data parallel: this layer's gradients are final
        self.saved = []
