"""Full-parameter fine-tuning on libvlb (BASELINE configs[4]).

What the reference gets with ``freeze_backbone=False, use_lora=False`` (src/litmodule/videollama2_vlb_litmodule.py:
86-99): ``requires_grad`` stays on for the whole ``Videollama2MistralForCausalLM`` except the vision tower, so
``loss.backward()`` differentiates the STC connector, ``embed_tokens``, all 32 decoder layers and the final norm
(``lm_head`` is a parameter too, but the loss never reads the logits: its gradient is ``None`` and AdamW skips it,
so it is neither held nor updated here).

MI355X layout: every weight gradient of a linear / 1x1 conv is the SAME TN MFMA GEMM as the forward,
``dW[N,K] = dy^T[N,M] . (x^T[K,M])^T`` on activations transposed once per use (``vlb_transpose_pad``: token axis
padded with zeros to the GEMM's K granule); dgrad uses the transposed weights, refreshed after every optimiser step.
All trained backbone tensors live, in their kernel layouts, in one flat store (``FlatBackbone``: bf16 weights = the
tensors the kernels read, bf16 gradients = what the wgrad GEMMs write, fp32 master + Adam moments) cut into one
segment per decoder layer plus a tail segment (connector, embeddings, final norm), so ``parallel.ShardedFlatState``
reduce-scatters layer chunks under the remaining backward and shards the 12 B/param optimiser state 1/world.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import ops
from ._lib import check, lib
from .flat import SEG_ALIGN
from .geometry import Geometry, VIDEO_TOKEN_ID

BF16 = torch.bfloat16
CONN_W = ("conv1", "dw", "se1", "se1b", "se2", "se2b", "conv3", "ds")        # single-tensor entries of a bottleneck block
CONN_LN = ("bn1", "bn2", "bn3", "dsbn")                                      # (weight, bias) pairs


def _stream():
    return torch.cuda.current_stream().cuda_stream


# fp8 path: quantisations produced by the kernel that writes the bf16 tensor (rmsnorm / SwiGLU) or from one read for both layouts
# (dy and W: row-wise + transposed).  VLB_FP8_FUSED_QUANT=0 runs the separate quantise passes instead (A/B; results identical).
FUSED_QUANT = os.environ.get("VLB_FP8_FUSED_QUANT", "1") == "1"


def _up(n, m):
    return (n + m - 1) // m * m


class FlatBackbone:
    """Flat buffers for every trained backbone tensor (kernel layouts).  Same attribute surface as
    ``flat.FlatTrainables`` (``numel, master, compute, grad, m, v, head_range, layer_ranges, offsets``) so the optimiser
    and ``parallel.ShardedFlatState`` treat both alike; ``grad`` is bf16 here.  ``head_range`` is the TAIL segment
    (connector + embeddings + final norm): its gradients are final last, so it is the segment reduced at the end."""

    grad_bf16 = True

    def __init__(self, g: Geometry, w):
        self.g = g
        slots = []        # (name, tensor, setter, segment)   segment -1 = tail

        def slot(name, holder, key, seg, idx=None):
            if idx is None:
                slots.append((name, holder[key], lambda t, h=holder, k=key: h.__setitem__(k, t), seg))
            else:
                def setter(t, h=holder, k=key, i=idx):
                    pair = list(h[k]); pair[i] = t; h[k] = tuple(pair)
                slots.append((name, holder[key][idx], setter, seg))
        for st, blocks in (("s1", w.s1), ("s2", w.s2)):
            for bi, blk in enumerate(blocks):
                for k in CONN_W:
                    if k in blk:
                        slot(f"mm_projector.{st}.b{bi + 1}.{k}", blk, k, -1)
                for k in CONN_LN:
                    if k in blk:
                        slot(f"mm_projector.{st}.b{bi + 1}.{k}.weight", blk, k, -1, 0)
                        slot(f"mm_projector.{st}.b{bi + 1}.{k}.bias", blk, k, -1, 1)
        wd = w.__dict__
        slot("mm_projector.sampler.weight", wd, "sampler_w", -1)
        slot("mm_projector.sampler.bias", wd, "sampler_b", -1)
        for k in ("ro0", "ro2"):
            slot(f"mm_projector.{k}.weight", wd, k, -1, 0)
            slot(f"mm_projector.{k}.bias", wd, k, -1, 1)
        slot("embed_tokens", wd, "embed", -1)
        slot("norm", wd, "final_norm", -1)
        for li, lw in enumerate(w.layers):
            for k in ("wqkv", "wo", "wgu", "wdown", "in_norm", "post_norm"):
                slot(f"layers.{li}.{k}", lw, k, li)
        offs, off, seg_start = {}, 0, 0
        self.layer_ranges = []
        for i, (name, t, _, seg) in enumerate(slots):
            offs[name] = (off, t.numel(), tuple(t.shape))
            off += _up(t.numel(), 8)
            if i + 1 == len(slots) or slots[i + 1][3] != seg:
                off = _up(off, SEG_ALIGN)
                if seg < 0:
                    self.head_range = (seg_start, off)
                else:
                    self.layer_ranges.append((seg_start, off))
                seg_start = off
        self.numel, self.offsets = off, offs
        dev = w.dev
        self.compute = torch.zeros(off, dtype=BF16, device=dev)
        self._setters = {}                              # name -> (setter, segment): release_layers() re-points the views
        for name, t, setter, seg in slots:
            o, k, shp = offs[name]
            v = self.compute[o:o + k].view(shp)
            v.copy_(t)
            setter(v)                                   # the kernels now read (and AdamW rewrites) the flat buffer
            self._setters[name] = (setter, seg)
        self.layers_released = False
        self.layer_grad_provider = None                 # FULL_SHARD: callable(layer) -> that layer's flat gradient buffer
        del slots
        torch.cuda.empty_cache()
        self.master = torch.empty(off, dtype=torch.float32, device=dev)
        step = 1 << 28
        for a in range(0, off, step):
            self.master[a:a + step].copy_(self.compute[a:a + step])
        self.grad = torch.zeros(off, dtype=BF16, device=dev)
        self.m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.v = torch.zeros(off, dtype=torch.float32, device=dev)

    LAYER_KEYS = ("wqkv", "wo", "wgu", "wdown", "in_norm", "post_norm")

    def view(self, buf, name):
        o, k, shp = self.offsets[name]
        if o + k > buf.numel():
            raise RuntimeError(f"{name}: this store holds the tail segment only (FULL_SHARD) - layer tensors are reached through "
                               f"parallel.ShardedFlatState.layer_weights / layer_grads / gather_full")
        return buf[o:o + k].view(shp)

    def g_(self, name):
        if self.layer_grad_provider is not None and name.startswith("layers."):
            li = int(name.split(".")[1])
            o, k, shp = self.offsets[name]
            o -= self.layer_ranges[li][0]
            return self.layer_grad_provider(li)[o:o + k].view(shp)
        return self.view(self.grad, name)

    def layer_views(self, buf, li: int) -> dict:
        """The six tensors of decoder layer li as views of a flat layer buffer (a gathered weight or gradient buffer)."""
        base, out = self.layer_ranges[li][0], {}
        for key in self.LAYER_KEYS:
            o, k, shp = self.offsets[f"layers.{li}.{key}"]
            out[key] = buf[o - base:o - base + k].view(shp)
        return out

    def release_layers(self):
        """FULL_SHARD (parallel.ShardedFlatState._enter_full_shard): keep the TAIL segment only.  ``compute`` and ``grad``
        shrink to it (the tail's kernel views are re-pointed), the full-size fp32 ``master`` goes, every layer view becomes
        None so that a stale use fails loudly."""
        T = self.head_range[1]
        tail = self.compute[:T].clone()
        for name, (setter, seg) in self._setters.items():
            o, k, shp = self.offsets[name]
            setter(tail[o:o + k].view(shp) if seg < 0 else None)
        self.compute, self.grad, self.master = tail, torch.zeros(T, dtype=BF16, device=tail.device), None
        self.layers_released = True


class FullFineTune:
    """Forward-with-save and explicit backward of connector + decoder for the full-parameter configuration."""

    def __init__(self, g: Geometry, backbone, device, fp8_gemm: bool = False):
        """``fp8_gemm``: run the decoder's forward and dgrad GEMMs on the block-scaled fp8 MFMA path (MX e4m3 with
        E8M0 scales per 32 K elements, ``vlb_gemm_mxfp8``): activations are quantised per use, the weights (and their
        transposes) once per optimiser step, the decoder's weight gradients quantise their transposed operands along the
        token axis; attention, norms, the connector and the optimiser stay as they are."""
        self.g, self.bb, self.w, self.dev = g, backbone, backbone.w, device
        backbone.trainable_decoder = True     # Backbone.enable_sharding refuses: these weights are optimiser state
        self.fp8 = bool(fp8_gemm)
        self.wq = {}                          # (layer, key) -> (uint8 e4m3 weights, uint8 scales)
        self.flat = FlatBackbone(g, self.w)
        self.grad_hook = None                 # callable(layer index): that layer's gradients are final (data parallel)
        self.shards = None                    # parallel.ShardedFlatState in FULL_SHARD mode (enter_full_shard)
        self._lw = self._lq = None            # weights (and fp8 quantisations) of the decoder layer being computed
        self._tb = {}
        # transposed copies for dgrad: the decoder's come from Weights(keep_transposed=True); the connector's are made here
        self.conn_t = {}
        self.refresh_transposed(decoder=self.fp8)

    # ------------------------------------------------------------------ FULL_SHARD (fsdp.yaml:11) for the decoder weights
    def enter_full_shard(self, shards):
        """Called by ``parallel.attach_data_parallel``: from here on the decoder layers' weights exist on this rank as 1/world
        slices only.  Per layer and pass ``_enter`` has the layer gathered (the next one already on its way), and derives what
        that pass needs on the spot: the W^T copies for dgrad (bf16 path; one set of buffers - the per-layer copies kept by
        ``Weights(keep_transposed=True)`` are released) or the MX-fp8 quantisations (row-wise for the forward, transposed for
        dgrad).  That is the same derivation work ``refresh_transposed`` does once per step for the replicated layout, moved
        to where the gathered layer is."""
        self.shards = shards
        self.flat.layer_grad_provider = shards.layer_grads
        self.bb.store = shards                          # evaluation forward (Backbone.layer_weights) reads through the same gathers
        self.bb.store_t = None
        for lw in self.w.layers:
            for k in ("wqkv_t", "wo_t", "wgu_t", "wdown_t"):
                if k in lw:
                    lw[k] = None
        self.wq = {}
        self._tpool = {}

    MATS = ("wqkv", "wo", "wgu", "wdown")

    def _enter(self, li: int, backward: bool) -> dict:
        """Weights of decoder layer li for this pass -> ``self._lw`` (and ``self._lq`` on the fp8 path)."""
        if self.shards is None:
            self._lw = self.w.layers[li]
            self._lq = {k: self.wq[(li, k)] for k in self.MATS + tuple(m + "_t" for m in self.MATS)} if self.fp8 else None
            return self._lw
        L = self.g.layers
        nxt = li - 1 if backward else min(li + 1, L - 1)          # (the last forward layer is the first of the backward)
        lw = self.flat.layer_views(self.shards.layer_weights(li, then=nxt if nxt >= 0 else None), li)
        lq = {} if self.fp8 else None
        for k in self.MATS:
            N, K = lw[k].shape
            if self.fp8:
                if backward:
                    q = self._pool(("qt", k), (K, N), torch.uint8), self._pool(("st", k), (K, N // 32), torch.uint8)
                    lq[k + "_t"] = ops.transpose_quantize_mxfp8(lw[k], q[0], q[1], N)
                else:
                    q = self._pool(("q", k), (N, K), torch.uint8), self._pool(("s", k), (N, K // 32), torch.uint8)
                    lq[k] = ops.quantize_mxfp8(lw[k], q[0], q[1])
            elif backward:
                t = self._pool(("t", k), (K, N), BF16)
                self._wt(lw[k], t)
                lw[k + "_t"] = t
        self._lw, self._lq = lw, lq
        return lw

    def _pool(self, key, shape, dtype):
        t = self._tpool.get(key)
        if t is None:
            t = self._tpool[key] = torch.empty(shape, dtype=dtype, device=self.dev)
        return t

    # ------------------------------------------------------------------ derived layouts
    def _conn_linears(self):
        for st, blocks in (("s1", self.w.s1), ("s2", self.w.s2)):
            for bi, blk in enumerate(blocks):
                for k in ("conv1", "se1", "se2", "conv3", "ds"):
                    if k in blk:
                        yield f"{st}.{bi}.{k}", blk[k]
        yield "sampler", self.w.sampler_w
        yield "ro0", self.w.ro0[0]
        yield "ro2", self.w.ro2[0]

    @staticmethod
    def _wt(w, w_t):
        """w_t[C, R] = w[R, C]^T with 16-byte accesses on both sides (R % 8 == 0 for every trained matrix)."""
        R, C = w.shape
        if R % 8 == 0:
            ops.transpose_pad(w, w_t, R)
        else:
            check(lib.vlb_transpose_bf16(w.data_ptr(), w_t.data_ptr(), R, C, _stream()), "vlb_transpose_bf16")

    def refresh_transposed(self, decoder=True):
        """After an optimiser step: W^T copies for the dgrad GEMMs (two passes over the trained weights)."""
        for name, wt in self._conn_linears():
            R, C = wt.shape
            if name not in self.conn_t:
                self.conn_t[name] = torch.empty(C, R, dtype=BF16, device=self.dev)
            self._wt(wt, self.conn_t[name])
        if self.shards is not None:           # FULL_SHARD: the decoder's derived layouts are made per gathered layer (_enter)
            decoder = False
        if decoder and not self.fp8:          # (the fp8 path quantises W^T straight from W below)
            for lw in self.w.layers:
                for k in ("wqkv", "wo", "wgu", "wdown"):
                    self._wt(lw[k], lw[k + "_t"])
        if self.fp8 and self.shards is None:  # quantised W for the forward, quantised W^T (straight from W, one pass) for dgrad
            for li, lw in enumerate(self.w.layers):
                for k in ("wqkv", "wo", "wgu", "wdown"):
                    N, K = lw[k].shape
                    old = self.wq.get((li, k + "_t"))
                    if old is None:
                        old = (torch.empty(K, N, dtype=torch.uint8, device=self.dev), torch.empty(K, N // 32, dtype=torch.uint8, device=self.dev))
                    if FUSED_QUANT and K % 64 == 0 and N % 32 == 0:         # both quantisations from ONE read of W (vlb_quantize_dual_mxfp8)
                        row = self.wq.get((li, k))
                        if row is None:
                            row = (torch.empty(N, K, dtype=torch.uint8, device=self.dev), torch.empty(N, K // 32, dtype=torch.uint8, device=self.dev))
                        ops.quantize_dual_mxfp8(lw[k], row[0], row[1], old[0], old[1], N)
                        self.wq[(li, k)], self.wq[(li, k + "_t")] = row, old
                    else:
                        self.wq[(li, k)] = ops.quantize_mxfp8(lw[k], *self.wq.get((li, k), (None, None)))
                        self.wq[(li, k + "_t")] = ops.transpose_quantize_mxfp8(lw[k], old[0], old[1], N)
        dw_f = getattr(self, "dw_flipped", None)
        if dw_f is None:
            dw_f = self.dw_flipped = {}
        for st, blocks in (("s1", self.w.s1), ("s2", self.w.s2)):
            for bi, blk in enumerate(blocks):
                dw_f[(st, bi)] = torch.flip(blk["dw"], dims=[0]).contiguous()      # dx of a depthwise 3x3 = the same conv, taps reversed

    # ------------------------------------------------------------------ weight gradients
    def _tbuf(self, tag, rows, cols):
        key = tag
        buf = self._tb.get(key)
        if buf is None or buf.shape[0] < rows or buf.shape[1] < cols:
            r0 = max(rows, buf.shape[0] if buf is not None else 0)
            c0 = max(cols, buf.shape[1] if buf is not None else 0)
            buf = self._tb[key] = torch.empty(r0, c0, dtype=BF16, device=self.dev)
        return buf

    def _transposed(self, x, tag, gran=64):
        """x [M, C] -> view [C, Mp] of x^T with the token axis zero-padded to a multiple of `gran` (the GEMM's K granule)."""
        M, C = x.shape
        Mp = _up(M, gran)
        buf = self._tbuf(tag, C, Mp)
        ops.transpose_pad(x, buf, Mp)
        return buf[:C, :Mp]

    def wgrad(self, dy, x, out, dyT=None, xT=None, fp8=False, dyTq=None):
        """out[N, K] (bf16 view into the flat gradient buffer) = dy[M,N]^T . x[M,K].  fp8: both transposed operands are
        quantised along the token axis (MX blocks of 32 tokens) and the product runs on the MX-fp8 MFMA GEMM; ``dyTq``:
        the transposed quantisation of dy when `_dual` already produced it."""
        if fp8 and out.shape[1] % 256 == 0:
            Mp = _up(dy.shape[0], 128)
            if dyTq is None:
                qa, sa = self._qbuf("dyTq", dy.shape[1], Mp)
                ops.transpose_quantize_mxfp8(dy, qa, sa, Mp)      # transpose and quantise in one pass (3 B/element)
            else:
                qa, sa = dyTq
            qb, sb = self._qbuf("xTq", x.shape[1], Mp)
            ops.transpose_quantize_mxfp8(x, qb, sb, Mp)
            ops.gemm_mxfp8(qa, sa, qb, sb, out=out)
            return
        dyT = self._transposed(dy, "dyT") if dyT is None else dyT
        xT = self._transposed(x, "xT") if xT is None else xT
        ops.gemm(dyT, xT, out=out)

    def _qbuf(self, tag, rows, cols):
        """Grow-only (uint8 e4m3 [rows, cols], uint8 scales [rows, cols/32]) views for quantised transposed operands."""
        need = rows * cols
        buf = self._tb.get(tag)
        if buf is None or buf[0].numel() < need:
            buf = self._tb[tag] = (torch.empty(need, dtype=torch.uint8, device=self.dev), torch.empty(need // 32, dtype=torch.uint8, device=self.dev))
        return buf[0][:need].view(rows, cols), buf[1][:need // 32].view(rows, cols // 32)

    def _lin(self, x, li, key, residual=None, xq=None):
        """x @ W^T (+ residual) for decoder weight `key` of layer li: bf16 MFMA GEMM, or the MX-fp8 one.  ``xq``: (e4m3, scales)
        of x when its producer already emitted them (rmsnorm_mxfp8 / swiglu_mxfp8 / quantize_dual_mxfp8)."""
        if self.fp8:
            xq, xs = ops.quantize_mxfp8(x) if xq is None else xq
            wq, ws = self._lq[key]
            return ops.gemm_mxfp8(xq, xs, wq, ws, residual=residual)
        return ops.gemm(x, self._lw[key], residual=residual)

    def _norm(self, x, wn):
        """(h, quantised h or None): RMSNorm; on the fp8 path the kernel emits the MX quantisation of h in the same pass."""
        if self.fp8 and FUSED_QUANT and x.shape[1] % 32 == 0 and x.shape[1] <= 4096:
            h, q, s = ops.rmsnorm_mxfp8(x, wn, self.g.rms_eps)
            return h, (q, s)
        return ops.rmsnorm(x, wn, self.g.rms_eps), None

    def _dual(self, dy):
        """Row-wise and transposed MX quantisation of a backward signal from one read (dgrad and wgrad operands of one linear);
        None when the shape does not fit the fused kernel (the callers then quantise separately)."""
        if not self.fp8 or not FUSED_QUANT or dy.shape[1] % 64 != 0:
            return None
        M, C = dy.shape
        Mp = _up(M, 128)
        q, s = self._qbuf("dyq", M, C)
        qt, st = self._qbuf("dyTq", C, Mp)
        ops.quantize_dual_mxfp8(dy, q, s, qt, st, Mp)
        return (q, s), (qt, st)

    # ------------------------------------------------------------------ connector (training forward + backward)
    def _block_fwd(self, x, blk, N, H, save):
        g, C = self.g, self.g.dim
        y1 = ops.gemm(x, blk["conv1"])
        a1 = ops.layernorm(y1, *blk["bn1"], g.proj_eps, act=ops.ACT_SILU)
        y2 = ops.dwconv3x3(a1, blk["dw"], N, H, H, C)
        a2 = ops.layernorm(y2, *blk["bn2"], g.proj_eps, act=ops.ACT_SILU)
        p = ops.se_pool(a2, N, H * H, C)
        q1 = ops.gemm(p, blk["se1"], bias=blk["se1b"])
        r1 = ops.act_fwd(q1, ops.ACT_SILU)
        s = ops.gemm(r1, blk["se2"], bias=blk["se2b"])
        a3 = ops.se_scale(a2, s, N, H * H, C)
        y3 = ops.gemm(a3, blk["conv3"])
        ysc, sc = None, x
        if "ds" in blk:
            ysc = ops.gemm(x, blk["ds"])
            sc = ops.layernorm(ysc, *blk["dsbn"], g.proj_eps)
        out = ops.layernorm(y3, *blk["bn3"], g.proj_eps, residual=sc, act=ops.ACT_SILU)
        save.append(dict(x=x, y1=y1, a1=a1, y2=y2, a2=a2, p=p, q1=q1, r1=r1, s=s, a3=a3, y3=y3, ysc=ysc, sc=sc, N=N, H=H))
        return out

    def connector_forward(self, feats, B):
        g, w = self.g, self.w
        sv = self.conn_saved = dict(s1=[], s2=[], B=B)
        x = feats
        for blk in w.s1:
            x = self._block_fwd(x, blk, B * g.num_frames, g.grid, sv["s1"])
        cols = ops.im2col3d(x, B, g.num_frames, g.grid, g.grid, g.dim)
        ypre = ops.gemm(cols, w.sampler_w, bias=w.sampler_b)
        x = ops.act_fwd(ypre, ops.ACT_SILU)
        sv.update(cols=cols, ypre=ypre)
        for blk in w.s2:
            x = self._block_fwd(x, blk, B * g.ds_frames, g.ds_grid, sv["s2"])
        t0 = ops.gemm(x, w.ro0[0], bias=w.ro0[1])
        t1 = ops.act_fwd(t0, ops.ACT_GELU)
        sv.update(ro_in=x, t0=t0, t1=t1)
        return ops.gemm(t1, w.ro2[0], bias=w.ro2[1])

    def _block_bwd(self, d_out, blk, st, bi, sv, need_dx):
        g, C, G = self.g, self.g.dim, self.flat.g_
        pre = f"mm_projector.{st}.b{bi + 1}"
        N, H = sv["N"], sv["H"]
        dy3, dsc = ops.layernorm_bwd(sv["y3"], *blk["bn3"], d_out, g.proj_eps, G(f"{pre}.bn3.weight"), G(f"{pre}.bn3.bias"),
                                     residual=sv["sc"], act=ops.ACT_SILU, want_dres=True)
        self.wgrad(dy3, sv["a3"], G(f"{pre}.conv3"))
        d_a3 = ops.gemm(dy3, self.conn_t[f"{st}.{bi}.conv3"])
        ds = ops.se_bwd_gate(sv["a2"], d_a3, sv["s"], N, H * H, C)
        self.wgrad(ds, sv["r1"], G(f"{pre}.se2"))
        ops.colsum(ds, G(f"{pre}.se2b"))
        d_r1 = ops.gemm(ds, self.conn_t[f"{st}.{bi}.se2"])
        d_q1 = ops.act_bwd(sv["q1"], d_r1, ops.ACT_SILU)
        self.wgrad(d_q1, sv["p"], G(f"{pre}.se1"))
        ops.colsum(d_q1, G(f"{pre}.se1b"))
        d_p = ops.gemm(d_q1, self.conn_t[f"{st}.{bi}.se1"])
        d_a2 = ops.se_bwd_x(d_a3, sv["s"], d_p, N, H * H, C)
        dy2, _ = ops.layernorm_bwd(sv["y2"], *blk["bn2"], d_a2, g.proj_eps, G(f"{pre}.bn2.weight"), G(f"{pre}.bn2.bias"), act=ops.ACT_SILU)
        ops.dwconv3x3_bwd_w(sv["a1"], dy2, N, H, H, C, G(f"{pre}.dw"))
        d_a1 = ops.dwconv3x3(dy2, self.dw_flipped[(st, bi)], N, H, H, C)
        dy1, _ = ops.layernorm_bwd(sv["y1"], *blk["bn1"], d_a1, g.proj_eps, G(f"{pre}.bn1.weight"), G(f"{pre}.bn1.bias"), act=ops.ACT_SILU)
        xT = self._transposed(sv["x"], "xT")
        self.wgrad(dy1, sv["x"], G(f"{pre}.conv1"), xT=xT)
        dysc = None
        if "ds" in blk:
            dysc, _ = ops.layernorm_bwd(sv["ysc"], *blk["dsbn"], dsc, g.proj_eps, G(f"{pre}.dsbn.weight"), G(f"{pre}.dsbn.bias"))
            self.wgrad(dysc, sv["x"], G(f"{pre}.ds"), xT=xT)
        if not need_dx:
            return None
        if dysc is not None:
            dx = ops.gemm(dy1, self.conn_t[f"{st}.{bi}.conv1"])
            return ops.gemm(dysc, self.conn_t[f"{st}.{bi}.ds"], residual=dx)
        return ops.gemm(dy1, self.conn_t[f"{st}.{bi}.conv1"], residual=dsc)

    def connector_backward(self, d_vid):
        """d_vid: gradient of the video tokens bf16 [B*vis_tokens, dim].  Fills the connector gradients."""
        g, w, sv, G = self.g, self.w, self.conn_saved, self.flat.g_
        B = sv["B"]
        self.wgrad(d_vid, sv["t1"], G("mm_projector.ro2.weight"))
        ops.colsum(d_vid, G("mm_projector.ro2.bias"))
        d_t1 = ops.gemm(d_vid, self.conn_t["ro2"])
        d_t0 = ops.act_bwd(sv["t0"], d_t1, ops.ACT_GELU)
        self.wgrad(d_t0, sv["ro_in"], G("mm_projector.ro0.weight"))
        ops.colsum(d_t0, G("mm_projector.ro0.bias"))
        dx = ops.gemm(d_t0, self.conn_t["ro0"])
        for bi in range(len(w.s2) - 1, -1, -1):
            dx = self._block_bwd(dx, w.s2[bi], "s2", bi, sv["s2"][bi], True)
        d_ypre = ops.act_bwd(sv["ypre"], dx, ops.ACT_SILU)
        self.wgrad(d_ypre, sv["cols"], G("mm_projector.sampler.weight"))
        ops.colsum(d_ypre, G("mm_projector.sampler.bias"))
        dcols = ops.gemm(d_ypre, self.conn_t["sampler"])
        dx = ops.col2im3d(dcols, B, g.num_frames, g.grid, g.grid, g.dim)
        for bi in range(len(w.s1) - 1, -1, -1):
            dx = self._block_bwd(dx, w.s1[bi], "s1", bi, sv["s1"][bi], bi > 0)       # the vision tower below is frozen
        self.conn_saved = None

    # ------------------------------------------------------------------ whole backbone, training forward
    def forward(self, vision_f32, ids, layout=None, ids_host=None):
        g, bb, w = self.g, self.bb, self.w
        B = vision_f32.shape[0]
        vid = self.connector_forward(bb.video_tokens(vision_f32, tower_only=True), B)     # tower frozen: nothing kept; may have run one step ahead
        x, key_mask = bb.splice(ids, vid, layout)
        self._ids_host = ids_host if ids_host is not None else ids.cpu()
        S = g.max_len
        pos = None if layout is None else layout.pos
        qd, kd = g.heads * g.head_dim, g.kv_heads * g.head_dim
        self.saved = []
        for li in range(g.layers):
            lw = self._enter(li, backward=False)
            h1, h1q = self._norm(x, lw["in_norm"])
            qkv = self._lin(h1, li, "wqkv", xq=h1q)
            ops.rope_(qkv, w.rope_cos, w.rope_sin, B, S, g.heads + g.kv_heads, g.head_dim, pos=pos)
            a, lse = ops.attention_fwd(qkv[:, :qd], qkv[:, qd:qd + kd], qkv[:, qd + kd:], B, S, g.heads, g.kv_heads, g.head_dim,
                                       True, g.head_dim ** -0.5, key_mask=key_mask, need_lse=True, layout=layout)
            x2 = self._lin(a, li, "wo", residual=x)
            h2, h2q = self._norm(x2, lw["post_norm"])
            gu = self._lin(h2, li, "wgu", xq=h2q)
            if self.fp8 and FUSED_QUANT and gu.shape[1] % 64 == 0:
                hh, hq, hs = ops.swiglu_mxfp8(gu)
                x3 = self._lin(hh, li, "wdown", residual=x2, xq=(hq, hs))
            else:
                hh = ops.swiglu(gu)
                x3 = self._lin(hh, li, "wdown", residual=x2)
            self.saved.append(dict(x=x, h1=h1, qkv=qkv, a=a, lse=lse, x2=x2, h2=h2, gu=gu, hh=hh))
            x = x3
        self.x_last, self.key_mask, self.B, self.layout = x, key_mask, B, layout
        return ops.rmsnorm(x, w.final_norm, g.rms_eps), key_mask

    # ------------------------------------------------------------------ backward
    def backward(self, dhidden):
        """dhidden: d loss / d (post-final-norm hidden) bf16 [rows, dim].  Fills every backbone gradient."""
        g, w, G = self.g, self.w, self.flat.g_
        B, S = self.B, g.max_len
        qd, kd = g.heads * g.head_dim, g.kv_heads * g.head_dim
        layout = self.layout
        pos = None if layout is None else layout.pos
        dx = ops.rmsnorm_bwd_full(self.x_last, w.final_norm, dhidden, g.rms_eps, G("norm"))
        delta = torch.empty(B, g.heads, S, dtype=torch.float32, device=self.dev)
        for li in range(g.layers - 1, -1, -1):
            lw, sv = self._enter(li, backward=True), self.saved[li]
            pre = f"layers.{li}"
            # MLP: x3 = x2 + down(silu(gate) * up)
            dq_ = self._dual(dx) or (None, None)               # dx quantised both ways from one read
            self.wgrad(dx, sv["hh"], G(f"{pre}.wdown"), fp8=self.fp8, dyTq=dq_[1])
            d_hh = self._lin(dx, li, "wdown_t", xq=dq_[0])
            d_gu = ops.swiglu_bwd(sv["gu"], d_hh)
            dq_ = self._dual(d_gu) or (None, None)
            self.wgrad(d_gu, sv["h2"], G(f"{pre}.wgu"), fp8=self.fp8, dyTq=dq_[1])
            d_h2 = self._lin(d_gu, li, "wgu_t", xq=dq_[0])
            dx2 = ops.rmsnorm_bwd_full(sv["x2"], lw["post_norm"], d_h2, g.rms_eps, G(f"{pre}.post_norm"), dx_in=dx)
            # attention: x2 = x + o(attn(rope(qkv(norm(x)))))
            dq_ = self._dual(dx2) or (None, None)
            self.wgrad(dx2, sv["a"], G(f"{pre}.wo"), fp8=self.fp8, dyTq=dq_[1])
            d_a = self._lin(dx2, li, "wo_t", xq=dq_[0])
            dqkv = ops.attention_bwd(sv["qkv"], qd, kd, sv["a"], d_a, sv["lse"], self.key_mask, B, S, g.heads, g.kv_heads,
                                     g.head_dim, True, g.head_dim ** -0.5, layout=layout, delta=delta)
            ops.rope_(dqkv, w.rope_cos, w.rope_sin, B, S, g.heads + g.kv_heads, g.head_dim, sign=-1, pos=pos)
            dq_ = self._dual(dqkv) or (None, None)
            self.wgrad(dqkv, sv["h1"], G(f"{pre}.wqkv"), fp8=self.fp8, dyTq=dq_[1])
            d_h1 = self._lin(dqkv, li, "wqkv_t", xq=dq_[0])
            dx = ops.rmsnorm_bwd_full(sv["x"], lw["in_norm"], d_h1, g.rms_eps, G(f"{pre}.in_norm"), dx_in=dx2)
            self.saved[li] = None
            if self.grad_hook is not None:
                self.grad_hook(li)
        self._splice_backward(dx)
        self.saved = []

    def _splice_backward(self, d_embeds):
        """d inputs_embeds -> d embed_tokens (text rows, summed per token id in a fixed order) and d video tokens."""
        g = self.g
        ids = self._ids_host.numpy() if torch.is_tensor(self._ids_host) else np.asarray(self._ids_host)
        B, L = ids.shape
        Nv, S = g.vis_tokens, g.max_len
        layout = self.layout
        lens = layout.lens if (layout is not None and layout.packed) else [S] * B
        cu = np.concatenate([[0], np.cumsum(lens)]) if (layout is not None and layout.packed) else np.arange(B + 1) * S
        tok_rows, tok_ids, vid_rows = [], [], []
        for b in range(B):
            vpos = int(np.nonzero(ids[b] == VIDEO_TOKEN_ID)[0][0])
            j = np.arange(L)
            p = np.where(j < vpos, j, j + Nv - 1)
            keep = (j != vpos) & (p < lens[b])
            tok_rows.append(cu[b] + p[keep])
            tok_ids.append(ids[b][keep])
            vid_rows.append(cu[b] + vpos + np.arange(Nv))
        tok_rows, tok_ids = np.concatenate(tok_rows), np.concatenate(tok_ids)
        order = np.lexsort((tok_rows, tok_ids))                      # by token id, then by row: fixed summation order
        tok_rows, tok_ids = tok_rows[order], tok_ids[order]
        uniq, first = np.unique(tok_ids, return_index=True)
        beg = np.concatenate([first, [len(tok_ids)]])
        dev = self.dev
        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
        dE = self.flat.g_("embed_tokens")
        dE.zero_()                                                    # untouched vocabulary rows have zero gradient
        ops.embed_grad(d_embeds, i32(uniq), i32(beg), i32(tok_rows), dE, g.dim)
        vr = np.concatenate(vid_rows)
        d_vid = torch.empty(B * Nv, g.dim, dtype=BF16, device=dev)
        n = B * Nv
        ops.embed_grad(d_embeds, i32(np.arange(n)), i32(np.arange(n + 1)), i32(vr), d_vid, g.dim)
        self.connector_backward(d_vid)

    # ------------------------------------------------------------------ state dict (upstream names)
    def state_dict(self, which: str = "master") -> dict:
        """Trained backbone tensors under their upstream names and layouts, on the host: the fp32 masters, or with
        ``which="grad"`` the bf16 gradients of the last backward (tests compare them with autograd by name)."""
        g, f = self.g, self.flat
        buf = getattr(f, which)
        if self.shards is not None and (buf is None or buf.numel() < f.numel):
            # FULL_SHARD: no standing full-size copy - gathered from the owned slices (a collective: every rank calls this;
            # "grad" is then the REDUCED gradient of the last step)
            buf = self.shards.gather_full(which)
        M = lambda n: f.view(buf, n).detach().float().cpu()
        qd, kd = g.heads * g.head_dim, g.kv_heads * g.head_dim
        sd = {"model.embed_tokens.weight": M("embed_tokens"), "model.norm.weight": M("norm")}
        for li in range(g.layers):
            p, q = f"model.layers.{li}", f"layers.{li}"
            wqkv, wgu = M(f"{q}.wqkv"), M(f"{q}.wgu")
            sd[f"{p}.self_attn.q_proj.weight"] = wqkv[:qd].clone()
            sd[f"{p}.self_attn.k_proj.weight"] = wqkv[qd:qd + kd].clone()
            sd[f"{p}.self_attn.v_proj.weight"] = wqkv[qd + kd:].clone()
            sd[f"{p}.self_attn.o_proj.weight"] = M(f"{q}.wo")
            sd[f"{p}.mlp.gate_proj.weight"] = wgu[:g.ff].clone()
            sd[f"{p}.mlp.up_proj.weight"] = wgu[g.ff:].clone()
            sd[f"{p}.mlp.down_proj.weight"] = M(f"{q}.wdown")
            sd[f"{p}.input_layernorm.weight"] = M(f"{q}.in_norm")
            sd[f"{p}.post_attention_layernorm.weight"] = M(f"{q}.post_norm")
        P = "model.mm_projector"
        for st, blocks in (("s1", self.w.s1), ("s2", self.w.s2)):
            for bi, blk in enumerate(blocks):
                q, n = f"{P}.{st}.b{bi + 1}", f"mm_projector.{st}.b{bi + 1}"
                C = blk["conv1"].shape[0]
                sd[f"{q}.conv1.conv.weight"] = M(f"{n}.conv1")[:, :, None, None].clone()
                sd[f"{q}.conv2.conv.weight"] = M(f"{n}.dw").t().reshape(C, 1, 3, 3).clone()
                sd[f"{q}.conv3.conv.weight"] = M(f"{n}.conv3")[:, :, None, None].clone()
                for k, u in (("se1", "se.fc1"), ("se2", "se.fc2")):
                    sd[f"{q}.{u}.weight"] = M(f"{n}.{k}")[:, :, None, None].clone()
                    sd[f"{q}.{u}.bias"] = M(f"{n}.{k}b")
                for k, u in (("bn1", "conv1.bn"), ("bn2", "conv2.bn"), ("bn3", "conv3.bn")):
                    sd[f"{q}.{u}.weight"], sd[f"{q}.{u}.bias"] = M(f"{n}.{k}.weight"), M(f"{n}.{k}.bias")
                if "ds" in blk:
                    sd[f"{q}.downsample.conv.weight"] = M(f"{n}.ds")[:, :, None, None].clone()
                    sd[f"{q}.downsample.bn.weight"], sd[f"{q}.downsample.bn.bias"] = M(f"{n}.dsbn.weight"), M(f"{n}.dsbn.bias")
        sw = M("mm_projector.sampler.weight")
        C = sw.shape[0]
        sd[f"{P}.sampler.0.weight"] = sw.view(C, 2, 2, 2, C).permute(0, 4, 1, 2, 3).contiguous()
        sd[f"{P}.sampler.0.bias"] = M("mm_projector.sampler.bias")
        for k, u in (("ro0", "readout.0"), ("ro2", "readout.2")):
            sd[f"{P}.{u}.weight"], sd[f"{P}.{u}.bias"] = M(f"mm_projector.{k}.weight"), M(f"mm_projector.{k}.bias")
        return sd
