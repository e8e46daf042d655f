"""Tensor-level wrappers over the libvlb C-ABI (torch is plumbing: device memory + streams).

Every function takes CUDA(=HIP) tensors, passes raw pointers and the current stream to the
HIP kernels and returns torch tensors.  There is no CPU path: a CPU tensor raises.
"""
from __future__ import annotations

import torch

from ._lib import check, lib

ACT_NONE, ACT_QUICK_GELU, ACT_GELU, ACT_SILU, ACT_SWIGLU_PAIR = 0, 1, 2, 3, 4
BF16 = torch.bfloat16


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _dev(t):
    if not t.is_cuda:
        raise RuntimeError("phantom_vlb_amd ops need GPU tensors (no CPU fallback)")
    return t


class _GemmProbe:
    """HIP-event pairs (on the launch stream) around every vlb_gemm_bf16 call of one (N,K) shape.  With ``family`` the
    pairs go around EVERY long-K call (K + K2 >= 4096: the four-wave kernel family - plain, gate/up + SwiGLU, masked-pair
    dgrads, each with its tail launches) and carry the call's FLOPs; that mode costs ~2 events per GEMM and is only used
    outside a timed region."""

    def __init__(self, N, K, family=False):
        self.N, self.K, self.M = N, K, 0
        self.pairs = []
        self.family = family
        self.calls = []                      # family mode: (kind, flops, e0, e1)

    def result(self):
        torch.cuda.synchronize()
        if not self.pairs:
            return 0.0, 0, (0, self.N, self.K)
        ms = [a.elapsed_time(b) for a, b in self.pairs]
        return sum(ms) / len(ms), len(ms), (self.M, self.N, self.K)

    def family_result(self):
        """{kind: (calls, total ms, total TFLOP)} over everything recorded."""
        torch.cuda.synchronize()
        out = {}
        for kind, flops, e0, e1 in self.calls:
            c = out.setdefault(kind, [0, 0.0, 0.0])
            c[0] += 1
            c[1] += e0.elapsed_time(e1)
            c[2] += flops / 1e12
        return out


_probe = None


def enable_gemm_probe(N, K, family=False):
    global _probe
    _probe = _GemmProbe(N, K, family)
    return _probe


def disable_gemm_probe():
    global _probe
    _probe = None


class _Timed:
    """Context for one GEMM call: the (N, K) probe and / or the family probe."""

    def __init__(self, kind, M, N, K, K2=0):
        pr = _probe
        self.e = None
        if pr is None:
            return
        shape = N == pr.N and K == pr.K and kind != "masked"
        fam = pr.family and K + K2 >= 4096
        if not (shape or fam):
            return
        self.e = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        if shape and not pr.family:
            pr.M = M
            pr.pairs.append(self.e)
        if fam:
            pr.calls.append((kind, 2.0 * M * N * (K + K2)) + self.e)

    def __enter__(self):
        if self.e:
            self.e[0].record()

    def __exit__(self, *a):
        if self.e:
            self.e[1].record()


_gemm_ws = {}
split_k_tails = True        # False: never pass the workspace (every output element then sums K in one fixed order
                            # whatever the row count - what the packed == dense bit-identity tests rely on)


def _gemm_workspace(device):
    """One scratch buffer per (device, stream) for the split-K tails of vlb_gemm_bf16_ws."""
    if not split_k_tails:
        return None
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _gemm_ws.get(key)
    if ws is None:
        ws = _gemm_ws[key] = torch.empty(lib.vlb_gemm_workspace_bytes(), dtype=torch.uint8, device=device)
    return ws


def gemm(a, w, bias=None, residual=None, act=ACT_NONE, a2=None, w2=None, out=None):
    """out[M,N] = act(a[M,K] @ w[N,K]^T + a2 @ w2^T + bias) + residual   (all bf16, fp32 accumulate)."""
    _dev(a)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and a.stride(1) == 1 and w.stride(1) == 1
    if out is None:
        out = torch.empty(M, N // 2 if act == ACT_SWIGLU_PAIR else N, dtype=BF16, device=a.device)
    K2 = 0
    if a2 is not None:
        K2 = a2.shape[1]
        assert w2.shape == (N, K2) and a2.shape[0] == M
    ws = _gemm_workspace(a.device) if K + K2 >= 4096 and M * N > 256 * 192 * 256 else None    # more than one wave of tiles
    with _Timed("plain", M, N, K, K2):
        check(lib.vlb_gemm_bf16_ws(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0),
                                   M, N, K, _p(bias), _p(residual), residual.stride(0) if residual is not None else 0, act,
                                   _p(a2), a2.stride(0) if a2 is not None else 0, _p(w2),
                                   w2.stride(0) if w2 is not None else 0, K2, _p(ws), ws.numel() if ws is not None else 0,
                                   _stream()), "vlb_gemm_bf16")
    return out


def gemm_swiglu_save(a, w_il, a2=None, w2_il=None):
    """(h, gu): h[M, N/2] = silu(gate) * up and gu[M, N] = [gate | up] pre-activations of a @ w_il^T + a2 @ w2_il^T, with
    w_il / w2_il rows interleaved by ``interleave_gate_up`` - the gate/up projection, SwiGLU and the activations the
    backward pass keeps, in one GEMM (MistralMLP.forward, modeling_mistral.py:169-170)."""
    _dev(a)
    M, K = a.shape
    N = w_il.shape[0]
    assert w_il.shape[1] == K and a.stride(1) == 1 and w_il.stride(1) == 1 and N % 32 == 0
    h = torch.empty(M, N // 2, dtype=BF16, device=a.device)
    gu = torch.empty(M, N, dtype=BF16, device=a.device)
    K2 = 0
    if a2 is not None:
        K2 = a2.shape[1]
        assert w2_il.shape == (N, K2) and a2.shape[0] == M
    ws = _gemm_workspace(a.device) if K + K2 >= 4096 and M * N > 256 * 192 * 256 else None
    with _Timed("gate_up+swiglu", M, N, K, K2):
        check(lib.vlb_gemm_swiglu_save(a.data_ptr(), a.stride(0), w_il.data_ptr(), w_il.stride(0), h.data_ptr(), h.stride(0),
                                       gu.data_ptr(), gu.stride(0), M, N, K, _p(a2), a2.stride(0) if a2 is not None else 0, _p(w2_il),
                                       w2_il.stride(0) if w2_il is not None else 0, K2, _p(ws), ws.numel() if ws is not None else 0,
                                       _stream()), "vlb_gemm_swiglu_save")
    return h, gu


def gemm_masked_pair_ok(M, N, K):
    """Shapes vlb_gemm_bf16_masked_pair accepts (the fused LoRA dx path); callers fall back to gemm + lora_dx_masked."""
    return N % 256 == 0 and K % 64 == 0 and K >= 128 and M * (N // 2) < 2 ** 32


def gemm_masked_pair(a, w, a2, w2, p, seed, out=None):
    """out[M,N] = a @ w^T + keep(m,n)/(1-p) * (a2[:, :64] @ w2[:, :64]^T): LoRA backward through dropout in one GEMM."""
    _dev(a)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and a2.shape[0] == M and w2.shape[0] == N and a2.shape[1] >= 64 and w2.shape[1] >= 64
    if out is None:
        out = torch.empty(M, N, dtype=BF16, device=a.device)
    ws = _gemm_workspace(a.device) if K + 64 >= 4096 and M * N > 256 * 192 * 256 else None
    with _Timed("masked", M, N, K, 64):
        check(lib.vlb_gemm_bf16_masked_pair_ws(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0),
                                               M, N, K, a2.data_ptr(), a2.stride(0), w2.data_ptr(), w2.stride(0), float(p),
                                               int(seed) & 0xFFFFFFFF, _p(ws), ws.numel() if ws is not None else 0, _stream()),
              "vlb_gemm_bf16_masked_pair")
    return out


def gemm_masked_pair_swiglu_bwd(dy, w_t, gu, a2, w2, p, seed, out=None):
    """d[gate | up] of the MLP from dy (grad of the down projection's output): the LoRA-through-dropout dgrad GEMM
    with the SwiGLU backward in its epilogue (d_h never hits memory).  gu: saved [M, 2*ff] activations."""
    _dev(dy)
    M, K = dy.shape
    ff = w_t.shape[0]
    assert w_t.shape[1] == K and gu.shape == (M, 2 * ff) and a2.shape[0] == M and w2.shape[0] == ff
    if out is None:
        out = torch.empty(M, 2 * ff, dtype=BF16, device=dy.device)
    ws = _gemm_workspace(dy.device) if K + 64 >= 4096 and M * ff > 256 * 192 * 256 else None
    with _Timed("masked", M, ff, K, 64):
        check(lib.vlb_gemm_masked_pair_swiglu_bwd(dy.data_ptr(), dy.stride(0), w_t.data_ptr(), w_t.stride(0), gu.data_ptr(), gu.stride(0),
                                                  out.data_ptr(), out.stride(0), M, ff, K, a2.data_ptr(), a2.stride(0), w2.data_ptr(),
                                                  w2.stride(0), float(p), int(seed) & 0xFFFFFFFF, _p(ws),
                                                  ws.numel() if ws is not None else 0, _stream()), "vlb_gemm_masked_pair_swiglu_bwd")
    return out


def interleave_gate_up(w_gate, w_up):
    """[gate; up] -> 16-row blocks [gate_0 | up_0 | gate_1 | up_1 | ...] for ACT_SWIGLU_PAIR."""
    ff, k = w_gate.shape
    assert ff % 16 == 0
    return torch.stack([w_gate.view(ff // 16, 16, k), w_up.view(ff // 16, 16, k)], 1).reshape(2 * ff, k).contiguous()


def transpose(x):
    R, C = x.shape
    out = torch.empty(C, R, dtype=BF16, device=x.device)
    check(lib.vlb_transpose_bf16(_dev(x).data_ptr(), out.data_ptr(), R, C, _stream()), "vlb_transpose_bf16")
    return out


class RowLayout:
    """How the token rows of B clips are laid out: dense ``[B, S]`` (``lens=None``) or packed - clip b
    owns rows ``cu[b]..cu[b+1]`` and its padded tail is simply not there (the flash-attn varlen layout
    the reference's decoder runs on, modeling_mistral.py ``_upad_input``).  ``lens`` are HOST ints, so
    building a layout never reads device memory."""

    def __init__(self, B, S, lens=None, device=None):
        self.B, self.S = int(B), int(S)
        if lens is None:
            self.cu = self.pos = self.lens = None
            self.rows, self.smax = self.B * self.S, self.S
            return
        lens = [int(x) for x in lens]
        if len(lens) != self.B or min(lens) < 1 or max(lens) > self.S:
            raise ValueError(f"RowLayout: lens {lens} do not fit B={B}, S={S}")
        cu = [0]
        for n in lens:
            cu.append(cu[-1] + n)
        self.lens, self.rows, self.smax = lens, cu[-1], max(lens)
        self.cu = torch.tensor(cu, dtype=torch.int32).to(device, non_blocking=True)
        self.pos = torch.empty(self.rows, dtype=torch.int32, device=device)      # filled by splice_embed

    @property
    def packed(self):
        return self.cu is not None

    def unpack(self, x, fill=0.0):
        """[rows, C] -> dense [B, S, C] (padding rows = fill); test / inspection helper."""
        if not self.packed:
            return x.view(self.B, self.S, -1)
        out = torch.full((self.B, self.S, x.shape[1]), fill, dtype=x.dtype, device=x.device)
        o = 0
        for b, n in enumerate(self.lens):
            out[b, :n] = x[o:o + n]
            o += n
        return out


def attention_fwd(q, k, v, B, S, Hq, Hkv, D, causal, scale, key_mask=None, need_lse=False, out=None, layout=None):
    """q/k/v: 2-D views [rows, H*D] (row stride = token stride, may be slices of a packed qkv buffer).
    ``layout`` (a packed RowLayout) overrides B/S: rows follow layout.cu, lse is [B,Hq,layout.smax]."""
    _dev(q)
    cu, rows = None, B * S
    if layout is not None and layout.packed:
        B, S, cu, rows = layout.B, layout.smax, layout.cu, layout.rows
    assert q.shape[0] == rows, f"attention_fwd: {q.shape[0]} rows, layout has {rows}"
    if out is None:
        out = torch.empty(rows, Hq * D, dtype=BF16, device=q.device)
    lse = torch.empty(B, Hq, S, dtype=torch.float32, device=q.device) if need_lse else None
    check(lib.vlb_attention_fwd(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
                                out.data_ptr(), out.stride(0), _p(lse), _p(key_mask), B, S, Hq, Hkv, D,
                                1 if causal else 0, float(scale), _p(cu), _stream()), "vlb_attention_fwd")
    return (out, lse) if need_lse else out


_ATTN_WS = [None]


def _attn_workspace(n_floats, device):
    """Grow-only fp32 scratch for vlb_attention_bwd (per-q-head dK/dV partials), shared by all layers."""
    ws = _ATTN_WS[0]
    if ws is None or ws.numel() < n_floats or ws.device != device:
        ws = _ATTN_WS[0] = torch.empty(n_floats, dtype=torch.float32, device=device)
    return ws


def attention_bwd(qkv, qd, kd, out, dout, lse, key_mask, B, S, Hq, Hkv, D, causal, scale, layout=None, delta=None,
                  dq_acc=None):
    """Backward through attention_fwd on a fused [rows, qd+2*kd] qkv buffer -> dqkv of the same shape."""
    cu, rows = None, B * S
    if layout is not None and layout.packed:
        B, S, cu, rows = layout.B, layout.smax, layout.cu, layout.rows
    assert qkv.shape[0] == rows and dout.shape[0] == rows
    if delta is None:
        delta = torch.empty(B, Hq, S, dtype=torch.float32, device=qkv.device)
    if dq_acc is None and Hq != Hkv:
        dq_acc = _attn_workspace(rows * qd, qkv.device)
    assert delta.numel() >= B * Hq * S and (dq_acc is None or dq_acc.numel() >= rows * qd)
    dqkv = torch.empty_like(qkv)
    check(lib.vlb_attention_bwd(qkv.data_ptr(), qkv.stride(0), qkv[:, qd:].data_ptr(), qkv.stride(0),
                                qkv[:, qd + kd:].data_ptr(), qkv.stride(0), out.data_ptr(), out.stride(0),
                                dout.data_ptr(), dout.stride(0), lse.data_ptr(), _p(key_mask),
                                dqkv.data_ptr(), dqkv.stride(0), dqkv[:, qd:].data_ptr(), dqkv.stride(0),
                                dqkv[:, qd + kd:].data_ptr(), dqkv.stride(0), delta.data_ptr(), _p(dq_acc),
                                B, S, Hq, Hkv, D, 1 if causal else 0, float(scale), _p(cu), rows, _stream()),
          "vlb_attention_bwd")
    return dqkv


def rmsnorm(x, w, eps, out=None):
    rows, dim = x.shape
    if out is None:
        out = torch.empty_like(x)
    check(lib.vlb_rmsnorm_fwd(_dev(x).data_ptr(), w.data_ptr(), out.data_ptr(), rows, dim, eps, _stream()), "vlb_rmsnorm_fwd")
    return out


def rmsnorm_bwd(x, w, dy, eps, dx_in=None, out=None):
    rows, dim = x.shape
    if out is None:
        out = torch.empty_like(x)
    check(lib.vlb_rmsnorm_bwd(_dev(x).data_ptr(), w.data_ptr(), dy.data_ptr(), _p(dx_in), out.data_ptr(), rows, dim, eps,
                              _stream()), "vlb_rmsnorm_bwd")
    return out


def layernorm(x, w, b, eps, residual=None, act=ACT_NONE, out=None):
    rows, dim = x.shape
    if out is None:
        out = torch.empty_like(x)
    check(lib.vlb_layernorm_fwd(_dev(x).data_ptr(), w.data_ptr(), b.data_ptr(), _p(residual), out.data_ptr(), rows, dim,
                                eps, act, _stream()), "vlb_layernorm_fwd")
    return out


def rope_(x, cos, sin, B, S, heads, D, sign=1, pos=None):
    """In place on the first heads*D columns of the 2-D view x [rows, >=heads*D]; row r is at position
    pos[r] (packed layout) or r % S."""
    rows = B * S if pos is None else pos.shape[0]
    assert x.shape[0] == rows and cos.shape[0] >= S
    check(lib.vlb_rope_inplace(_dev(x).data_ptr(), x.stride(0), cos.data_ptr(), sin.data_ptr(), rows, S, heads, D, sign,
                               _p(pos), _stream()), "vlb_rope_inplace")
    return x


def swiglu(gu, out=None):
    rows, ff2 = gu.shape
    if out is None:
        out = torch.empty(rows, ff2 // 2, dtype=BF16, device=gu.device)
    check(lib.vlb_swiglu_fwd(_dev(gu).data_ptr(), out.data_ptr(), rows, ff2 // 2, _stream()), "vlb_swiglu_fwd")
    return out


def swiglu_bwd(gu, dout, out=None):
    rows, ff2 = gu.shape
    if out is None:
        out = torch.empty_like(gu)
    check(lib.vlb_swiglu_bwd(_dev(gu).data_ptr(), dout.data_ptr(), out.data_ptr(), rows, ff2 // 2, _stream()), "vlb_swiglu_bwd")
    return out


def add(a, b, out=None):
    if out is None:
        out = torch.empty_like(a)
    check(lib.vlb_add_bf16(_dev(a).data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "vlb_add_bf16")
    return out


def patchify(vision_f32, P, Kpad):
    N, C, H, W = vision_f32.shape
    assert C == 3 and vision_f32.dtype == torch.float32 and vision_f32.is_contiguous()
    out = torch.empty(N * (H // P) * (W // P), Kpad, dtype=BF16, device=vision_f32.device)
    check(lib.vlb_patchify(_dev(vision_f32).data_ptr(), out.data_ptr(), N, H, W, P, Kpad, _stream()), "vlb_patchify")
    return out


def vit_assemble(patch_emb, cls, pos, N, G, D):
    out = torch.empty(N * (G + 1), D, dtype=BF16, device=patch_emb.device)
    check(lib.vlb_vit_assemble(_dev(patch_emb).data_ptr(), cls.data_ptr(), pos.data_ptr(), out.data_ptr(), N, G, D,
                               _stream()), "vlb_vit_assemble")
    return out


def drop_cls(tokens, N, G, D):
    out = torch.empty(N * G, D, dtype=BF16, device=tokens.device)
    check(lib.vlb_drop_cls(_dev(tokens).data_ptr(), out.data_ptr(), N, G, D, _stream()), "vlb_drop_cls")
    return out


def dwconv3x3(x, w9, N, H, W, C):
    out = torch.empty_like(x)
    check(lib.vlb_dwconv3x3(_dev(x).data_ptr(), w9.data_ptr(), out.data_ptr(), N, H, W, C, _stream()), "vlb_dwconv3x3")
    return out


def se_pool(x, N, HW, C):
    out = torch.empty(N, C, dtype=BF16, device=x.device)
    check(lib.vlb_se_pool(_dev(x).data_ptr(), out.data_ptr(), N, HW, C, _stream()), "vlb_se_pool")
    return out


def se_scale(x, gate, N, HW, C, out=None):
    if out is None:
        out = torch.empty_like(x)
    check(lib.vlb_se_scale(_dev(x).data_ptr(), gate.data_ptr(), out.data_ptr(), N, HW, C, _stream()), "vlb_se_scale")
    return out


def im2col3d(x, B, T, H, W, C):
    T2, H2, W2 = T // 2 + 1, H // 2 + 1, W // 2 + 1
    out = torch.empty(B * T2 * H2 * W2, 8 * C, dtype=BF16, device=x.device)
    check(lib.vlb_im2col3d_k2s2p1(_dev(x).data_ptr(), out.data_ptr(), B, T, H, W, C, _stream()), "vlb_im2col3d_k2s2p1")
    return out


def splice_embed(ids, embed_w, video_tokens, Nv, video_id, err_flag, layout=None):
    """-> (embeds [rows, D], key_mask).  Dense: rows = B*S, mask [B,S].  With a packed ``layout`` only
    each clip's first lens[b] tokens are emitted (mask [rows]) and layout.pos receives their positions."""
    B, L = ids.shape
    D = embed_w.shape[1]
    S = L - 1 + Nv
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    packed = layout is not None and layout.packed
    if packed:
        assert layout.B == B and layout.S == S
    rows = layout.rows if packed else B * S
    out = torch.empty(rows, D, dtype=BF16, device=ids.device)
    mask = torch.empty((rows,) if packed else (B, S), dtype=torch.uint8, device=ids.device)
    check(lib.vlb_splice_embed(_dev(ids).data_ptr(), embed_w.data_ptr(), video_tokens.data_ptr(), out.data_ptr(),
                               mask.data_ptr(), err_flag.data_ptr(), B, L, Nv, D, video_id, embed_w.shape[0],
                               _p(layout.cu) if packed else None, _p(layout.pos) if packed else None,
                               _stream()), "vlb_splice_embed")
    return out, mask


def weight_mask(padvals, vis_w, lang_w, tokens_per_frame, S, round_bf16=True):
    B, F = vis_w.shape
    assert padvals.dtype == torch.int64 and vis_w.dtype == torch.float64 and lang_w.dtype == torch.float64
    out = torch.empty(B, S, dtype=torch.float32, device=padvals.device)
    check(lib.vlb_weight_mask(_dev(padvals).contiguous().data_ptr(), vis_w.contiguous().data_ptr(),
                              lang_w.contiguous().data_ptr(), out.data_ptr(), B, F, lang_w.shape[1], tokens_per_frame, S,
                              1 if round_bf16 else 0, _stream()), "vlb_weight_mask")
    return out


def dropout_keep_scale(B, E, p, seed, device):
    """fp32 [B,E] of keep/(1-p): the head's nn.Dropout(p) mask as a counter-based hash of (seed, position)."""
    out = torch.empty(B, E, dtype=torch.float32, device=device)
    check(lib.vlb_dropout_keep_scale(_dev(out).data_ptr(), out.numel(), float(p), int(seed) & 0xFFFFFFFF, _stream()),
          "vlb_dropout_keep_scale")
    return out


def profile_marker():
    """An empty launch named vlb_profile_marker_kernel on the current stream (bench.py brackets its timed region)."""
    check(lib.vlb_profile_marker(_stream()), "vlb_profile_marker")


def cast_bf16(x_f32):
    out = torch.empty(x_f32.shape, dtype=BF16, device=x_f32.device)
    check(lib.vlb_cast_f32_to_bf16(_dev(x_f32).data_ptr(), out.data_ptr(), x_f32.numel(), _stream()), "vlb_cast_f32_to_bf16")
    return out


# ---------------------------------------------------------------------------------------------------------------
# full-parameter fine-tuning pieces (csrc/train.hip)
_ws_cache = {}


def _ws(n_floats, device, tag="norm"):
    """Grow-only fp32 scratch per (device, tag) for the fixed-order column reductions."""
    key = (device.index, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < n_floats:
        buf = _ws_cache[key] = torch.empty(int(n_floats), dtype=torch.float32, device=device)
    return buf


def transpose_pad(x, out, rows_pad):
    """out[C, :rows_pad] = x[R, C]^T, columns R..rows_pad-1 zeroed (x may be a column slice: row stride respected)."""
    R, C = x.shape
    assert out.shape[0] >= C and out.stride(1) == 1 and x.stride(1) == 1 and rows_pad % 8 == 0 and out.shape[1] >= rows_pad
    check(lib.vlb_transpose_pad(_dev(x).data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), R, C, rows_pad, _stream()),
          "vlb_transpose_pad")
    return out


def rmsnorm_bwd_dw(x, dy, eps, out):
    rows, dim = x.shape
    ws = _ws(lib.vlb_norm_bwd_ws_floats(rows, dim), x.device)
    check(lib.vlb_rmsnorm_bwd_dw(_dev(x).data_ptr(), dy.data_ptr(), out.data_ptr(), ws.data_ptr(), rows, dim, eps, _stream()),
          "vlb_rmsnorm_bwd_dw")
    return out


def rmsnorm_bwd_full(x, w, dy, eps, dw_out, dx_in=None):
    """RMSNorm backward in one sweep: returns dx (+ dx_in), writes d gamma (bf16) into dw_out."""
    rows, dim = x.shape
    ws = _ws(lib.vlb_rmsnorm_bwd_full_ws_floats(rows, dim), x.device, "rmsfull")
    dx = torch.empty_like(x)
    check(lib.vlb_rmsnorm_bwd_full(_dev(x).data_ptr(), w.data_ptr(), dy.data_ptr(), _p(dx_in), dx.data_ptr(), dw_out.data_ptr(), ws.data_ptr(),
                                   rows, dim, eps, _stream()), "vlb_rmsnorm_bwd_full")
    return dx


def layernorm_bwd(x, w, b, dy, eps, dw_out, db_out, residual=None, act=ACT_NONE, want_dres=False):
    """Backward of ``layernorm(x, w, b, eps, residual, act)``: returns (dx, d residual or None); dw / db written (bf16)."""
    rows, dim = x.shape
    ws = _ws(lib.vlb_norm_bwd_ws_floats(rows, dim), x.device)
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    check(lib.vlb_layernorm_bwd(_dev(x).data_ptr(), w.data_ptr(), b.data_ptr(), _p(residual), dy.data_ptr(), dx.data_ptr(), _p(dres),
                                dw_out.data_ptr(), db_out.data_ptr(), ws.data_ptr(), rows, dim, eps, act, _stream()), "vlb_layernorm_bwd")
    return dx, dres


def act_fwd(x, act, out=None):
    out = torch.empty_like(x) if out is None else out
    check(lib.vlb_act_fwd(_dev(x).data_ptr(), out.data_ptr(), x.numel(), act, _stream()), "vlb_act_fwd")
    return out


def act_bwd(x, dy, act, out=None):
    out = torch.empty_like(x) if out is None else out
    check(lib.vlb_act_bwd(_dev(x).data_ptr(), dy.data_ptr(), out.data_ptr(), x.numel(), act, _stream()), "vlb_act_bwd")
    return out


def colsum(x, out):
    rows, dim = x.shape
    ws = _ws(lib.vlb_colsum_ws_floats(rows, dim), x.device)
    check(lib.vlb_colsum(_dev(x).data_ptr(), x.stride(0), out.data_ptr(), ws.data_ptr(), rows, dim, _stream()), "vlb_colsum")
    return out


def embed_grad(d_embeds, tok, beg, rows, out, D):
    """out[tok[j]] = sum of d_embeds[rows[beg[j]:beg[j+1]]] (int32 device lists built on the host)."""
    check(lib.vlb_embed_grad(_dev(d_embeds).data_ptr(), d_embeds.stride(0), tok.data_ptr(), beg.data_ptr(), rows.data_ptr(),
                             tok.numel(), out.data_ptr(), D, _stream()), "vlb_embed_grad")
    return out


def dwconv3x3_bwd_w(x, dy, N, H, W, C, out):
    ws = _ws(lib.vlb_dwconv3x3_bwd_w_ws_floats(N, H, C), x.device, "dw")
    check(lib.vlb_dwconv3x3_bwd_w(_dev(x).data_ptr(), dy.data_ptr(), out.data_ptr(), ws.data_ptr(), N, H, W, C, _stream()),
          "vlb_dwconv3x3_bwd_w")
    return out


def se_bwd_gate(x, dy, s, N, HW, C):
    ds = torch.empty(N, C, dtype=BF16, device=x.device)
    check(lib.vlb_se_bwd_gate(_dev(x).data_ptr(), dy.data_ptr(), s.data_ptr(), ds.data_ptr(), N, HW, C, _stream()), "vlb_se_bwd_gate")
    return ds


def se_bwd_x(dy, s, dpool, N, HW, C):
    dx = torch.empty_like(dy)
    check(lib.vlb_se_bwd_x(_dev(dy).data_ptr(), s.data_ptr(), _p(dpool), dx.data_ptr(), N, HW, C, _stream()), "vlb_se_bwd_x")
    return dx


def col2im3d(dcols, B, T, H, W, C):
    dx = torch.empty(B * T * H * W, C, dtype=BF16, device=dcols.device)
    check(lib.vlb_col2im3d_k2s2p1(_dev(dcols).data_ptr(), dx.data_ptr(), B, T, H, W, C, _stream()), "vlb_col2im3d_k2s2p1")
    return dx


# ---------------------------------------------------------------------------------------------------------------
# block-scaled fp8 (MX e4m3) GEMM path (csrc/gemm_fp8.hip)
def quantize_mxfp8(x, q=None, s=None):
    """bf16 [rows, K] -> (uint8 e4m3 [rows, K], uint8 E8M0 scales [rows, K/32]); x may be a column slice."""
    rows, K = x.shape
    assert K % 32 == 0 and x.stride(1) == 1
    q = torch.empty(rows, K, dtype=torch.uint8, device=x.device) if q is None else q
    s = torch.empty(rows, K // 32, dtype=torch.uint8, device=x.device) if s is None else s
    check(lib.vlb_quantize_mxfp8(_dev(x).data_ptr(), x.stride(0), q.data_ptr(), q.stride(0), s.data_ptr(), s.stride(0), rows, K, _stream()),
          "vlb_quantize_mxfp8")
    return q, s


def transpose_quantize_mxfp8(x, q, s, rows_pad):
    """(q [C, rows_pad] e4m3, s [C, rows_pad/32]) = MX quantisation of x[R, C]^T along x's row axis, one pass."""
    R, C = x.shape
    assert q.shape == (C, rows_pad) and s.shape == (C, rows_pad // 32) and x.stride(1) == 1
    check(lib.vlb_transpose_quantize_mxfp8(_dev(x).data_ptr(), x.stride(0), q.data_ptr(), q.stride(0), s.data_ptr(), s.stride(0), R, C,
                                           rows_pad, _stream()), "vlb_transpose_quantize_mxfp8")
    return q, s


def quantize_dual_mxfp8(x, q, s, qt, st, rows_pad):
    """Row-wise (q [R, C], s [R, C/32]) and transposed (qt [C, rows_pad], st [C, rows_pad/32]) MX quantisation of x[R, C] from one
    read of x; both bit-identical to quantize_mxfp8 / transpose_quantize_mxfp8."""
    R, C = x.shape
    assert x.stride(1) == 1 and q.shape == (R, C) and s.shape == (R, C // 32) and qt.shape == (C, rows_pad) and st.shape == (C, rows_pad // 32)
    check(lib.vlb_quantize_dual_mxfp8(_dev(x).data_ptr(), x.stride(0), q.data_ptr(), q.stride(0), s.data_ptr(), s.stride(0), qt.data_ptr(),
                                      qt.stride(0), st.data_ptr(), st.stride(0), R, C, rows_pad, _stream()), "vlb_quantize_dual_mxfp8")
    return q, s, qt, st


def rmsnorm_mxfp8(x, w, eps, q=None, s=None):
    """(y, q, s): RMSNorm plus the MX-fp8 quantisation of y in the same pass."""
    rows, dim = x.shape
    y = torch.empty_like(x)
    q = torch.empty(rows, dim, dtype=torch.uint8, device=x.device) if q is None else q
    s = torch.empty(rows, dim // 32, dtype=torch.uint8, device=x.device) if s is None else s
    check(lib.vlb_rmsnorm_fwd_mxfp8(_dev(x).data_ptr(), w.data_ptr(), y.data_ptr(), q.data_ptr(), q.stride(0), s.data_ptr(), s.stride(0),
                                    rows, dim, eps, _stream()), "vlb_rmsnorm_fwd_mxfp8")
    return y, q, s


def swiglu_mxfp8(gu, q=None, s=None):
    """(h, q, s): silu(gate) * up plus the MX-fp8 quantisation of h in the same pass."""
    rows, ff2 = gu.shape
    ff = ff2 // 2
    out = torch.empty(rows, ff, dtype=BF16, device=gu.device)
    q = torch.empty(rows, ff, dtype=torch.uint8, device=gu.device) if q is None else q
    s = torch.empty(rows, ff // 32, dtype=torch.uint8, device=gu.device) if s is None else s
    check(lib.vlb_swiglu_fwd_mxfp8(_dev(gu).data_ptr(), out.data_ptr(), q.data_ptr(), q.stride(0), s.data_ptr(), s.stride(0), rows, ff,
                                   _stream()), "vlb_swiglu_fwd_mxfp8")
    return out, q, s


def gemm_mxfp8(aq, sa, wq, sw, residual=None, out=None):
    """out[M,N] bf16 = dequant(aq, sa) @ dequant(wq, sw)^T + residual."""
    M, K = aq.shape
    N = wq.shape[0]
    assert wq.shape[1] == K and sa.shape == (M, K // 32) and sw.shape == (N, K // 32)
    if out is None:
        out = torch.empty(M, N, dtype=BF16, device=aq.device)
    timed = _probe is not None and N == _probe.N and K == _probe.K
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.vlb_gemm_mxfp8(_dev(aq).data_ptr(), aq.stride(0), sa.data_ptr(), sa.stride(0), wq.data_ptr(), wq.stride(0), sw.data_ptr(),
                             sw.stride(0), out.data_ptr(), out.stride(0), M, N, K, _p(residual),
                             residual.stride(0) if residual is not None else 0, _stream()), "vlb_gemm_mxfp8")
    if timed:
        e1.record()
        _probe.M = M
        _probe.pairs.append((e0, e1))
    return out
