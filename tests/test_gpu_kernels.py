"""Parity of every libvlb kernel against the CPU oracle / fp32 torch restatements (-m gpu).

All calls go through the C-ABI (phantom_vlb_amd.ops -> ctypes -> libvlb.so).  Tolerances are
relative to the reference's max magnitude: bf16 storage gives ~4e-3 per rounding; 1e-2 is the
bar for single kernels with bf16 outputs, 1e-4 for fp32 outputs.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _r(*shape, dev, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(BF).to(dev)


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [
    (512, 256, 128),      # 256x256 tile kernel
    (1000, 512, 192),     # M tail on the tile kernel (1000 = 3*256 + 232)
    (300, 128, 64),       # 256x128 kernel with M tail
    (1184, 384, 128),     # N % 256 != 0 -> 256x128
    (37, 100, 72),        # generic kernel, ragged everything
    (5, 128, 4096),       # skinny M -> generic
    (2048, 1024, 640),    # padded patch-embed K
])
def test_gemm_shapes(dev, M, N, K):
    from phantom_vlb_amd import ops
    a, w = _r(M, K, dev=dev), _r(N, K, dev=dev, scale=0.1)
    out = ops.gemm(a, w)
    ref = a.float() @ w.float().t()
    assert rel_err(out, ref) < 6e-3


@pytest.mark.parametrize("M,N,K,K2", [
    (600, 512, 4096, 0),      # long K -> four-wave 128x128-block kernel, M tail
    (600, 256, 4096, 64),     # ... with the LoRA operand pair appended to the K loop
    (4500, 4096, 128, 0),     # 288 tiles = 1.125 waves -> tail split (ping-pong kernel + 256x128 halves)
    (4500, 4096, 4096, 64),   # tail split behind the four-wave kernel, second operand pair
    (5861, 4096, 4096, 0),    # 368 tiles of 256 rows -> the cost model picks 192-row tiles (496 = 2 waves)
    (3000, 6144, 4096, 64),   # 192-row tiles (16x24 = 384 tiles = 256 + 128 re-cut into 192x128 halves)
])
def test_gemm_long_k_and_tail_split(dev, M, N, K, K2):
    from phantom_vlb_amd import ops
    a, w = _r(M, K, dev=dev), _r(N, K, dev=dev, scale=0.05)
    bias, res = _r(N, dev=dev), _r(M, N, dev=dev)
    a2 = w2 = None
    ref = a.float() @ w.float().t()
    if K2:
        a2, w2 = _r(M, K2, dev=dev, seed=3), _r(N, K2, dev=dev, scale=0.05, seed=4)
        ref = ref + a2.float() @ w2.float().t()
    out = ops.gemm(a, w, bias=bias, residual=res, act=ops.ACT_SILU, a2=a2, w2=w2)
    y = F.silu(ref + bias.float()) + res.float()
    assert rel_err(out, y) < 6e-3
    # a row's result must not depend on how many rows the call has (packed layouts rely on it): the same
    # rows through a differently tiled launch are bit-identical
    # (with the split-K tails off: they change the fp32 summation order of the tiles of a partial wave)
    ops.split_k_tails = False
    try:
        full = ops.gemm(a, w, bias=bias, residual=res, act=ops.ACT_SILU, a2=a2, w2=w2)
        half = ops.gemm(a[: M // 2 + 3], w, bias=bias, residual=res[: M // 2 + 3], act=ops.ACT_SILU,
                        a2=None if a2 is None else a2[: M // 2 + 3], w2=w2)
    finally:
        ops.split_k_tails = True
    assert torch.equal(half, full[: M // 2 + 3])


def test_gemm_swiglu_pair_long_k_tail_split(dev):
    from phantom_vlb_amd import ops
    M, ff, K = 4500, 2048, 4096
    a = _r(M, K, dev=dev)
    wg, wu = _r(ff, K, dev=dev, scale=0.03, seed=1), _r(ff, K, dev=dev, scale=0.03, seed=2)
    out = ops.gemm(a, ops.interleave_gate_up(wg, wu), act=ops.ACT_SWIGLU_PAIR)
    ref = F.silu(a.float() @ wg.float().t()) * (a.float() @ wu.float().t())
    assert rel_err(out, ref) < 8e-3


@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_gemm_epilogue(dev, act):
    from phantom_vlb_amd import ops
    M, N, K = 520, 256, 128
    a, w = _r(M, K, dev=dev), _r(N, K, dev=dev, scale=0.1)
    bias, res = _r(N, dev=dev), _r(M, N, dev=dev)
    out = ops.gemm(a, w, bias=bias, residual=res, act=act)
    y = a.float() @ w.float().t() + bias.float()
    y = [lambda t: t, lambda t: t * torch.sigmoid(1.702 * t), F.gelu, F.silu][act](y) + res.float()
    assert rel_err(out, y) < 6e-3
    # same through the generic kernel (odd N)
    out2 = ops.gemm(a, w[:250], bias=bias[:250].contiguous(), residual=res[:, :250].contiguous(), act=act)
    assert rel_err(out2, y[:, :250]) < 6e-3


@pytest.mark.parametrize("M,N,K,K2", [(768, 512, 256, 64), (100, 96, 64, 16)])
def test_gemm_second_operand_pair(dev, M, N, K, K2):
    """LoRA form: x W^T + (s x A^T) B^T accumulated in one launch."""
    from phantom_vlb_amd import ops
    a, w = _r(M, K, dev=dev), _r(N, K, dev=dev, scale=0.1)
    a2, w2 = _r(M, K2, dev=dev), _r(N, K2, dev=dev, scale=0.1)
    out = ops.gemm(a, w, a2=a2, w2=w2)
    ref = a.float() @ w.float().t() + a2.float() @ w2.float().t()
    assert rel_err(out, ref) < 6e-3


@pytest.mark.parametrize("M,ff,K", [(512, 256, 128), (300, 128, 64), (40, 48, 64)])
def test_gemm_swiglu_pair_epilogue(dev, M, ff, K):
    """gate/up GEMM with MistralMLP's act_fn(gate)*up fused (interleaved weight rows), tile + generic kernels."""
    from phantom_vlb_amd import ops
    a = _r(M, K, dev=dev)
    wg, wu = _r(ff, K, dev=dev, scale=0.1, seed=1), _r(ff, K, dev=dev, scale=0.1, seed=2)
    out = ops.gemm(a, ops.interleave_gate_up(wg, wu), act=ops.ACT_SWIGLU_PAIR)
    assert out.shape == (M, ff)
    ref = F.silu(a.float() @ wg.float().t()) * (a.float() @ wu.float().t())
    assert rel_err(out, ref) < 8e-3
    # identical to the unfused pair of kernels up to one bf16 rounding of the pre-activations
    unfused = ops.swiglu(ops.gemm(a, torch.cat([wg, wu], 0)))
    assert rel_err(out, unfused.float()) < 1.5e-2


@pytest.mark.parametrize("M,ff,K,K2", [(40, 48, 64, 16),           # generic kernel
                                        (512, 256, 128, 64),        # 8-wave tile kernel
                                        (1000, 512, 4096, 64),      # four-wave kernel, M tail
                                        (5015, 2048, 4096, 64)])    # four-wave kernel + split-K tail (reduce-kernel epilogue)
def test_gemm_swiglu_save(dev, M, ff, K, K2):
    """vlb_gemm_swiglu_save: the adapted gate/up projection of a LoRA forward in one GEMM - h = silu(gate)*up AND the saved
    [gate | up] pre-activations (plain column order) - against the plain GEMM (+ second operand pair) and fp32 torch."""
    from phantom_vlb_amd import ops
    a, t = _r(M, K, dev=dev), _r(M, K2, dev=dev, seed=3)
    wg, wu = _r(ff, K, dev=dev, scale=0.05, seed=1), _r(ff, K, dev=dev, scale=0.05, seed=2)
    bg, bu = _r(ff, K2, dev=dev, scale=0.05, seed=4), _r(ff, K2, dev=dev, scale=0.05, seed=5)
    h, gu = ops.gemm_swiglu_save(a, ops.interleave_gate_up(wg, wu), a2=t, w2_il=ops.interleave_gate_up(bg, bu))
    assert h.shape == (M, ff) and gu.shape == (M, 2 * ff)
    plain = ops.gemm(a, torch.cat([wg, wu], 0), a2=t, w2=torch.cat([bg, bu], 0))
    assert rel_err(gu, plain.float()) < 4e-3                  # same contraction; only the split-K grouping of tail tiles may differ
    g32 = a.float() @ wg.float().t() + t.float() @ bg.float().t()
    u32 = a.float() @ wu.float().t() + t.float() @ bu.float().t()
    assert rel_err(gu, torch.cat([g32, u32], 1)) < 6e-3
    assert rel_err(h, F.silu(g32) * u32) < 8e-3
    # without the second pair and without saving it is the frozen path's epilogue
    h0, gu0 = ops.gemm_swiglu_save(a, ops.interleave_gate_up(wg, wu))
    assert rel_err(h0, ops.gemm(a, ops.interleave_gate_up(wg, wu), act=ops.ACT_SWIGLU_PAIR).float()) < 4e-3
    assert rel_err(gu0, ops.gemm(a, torch.cat([wg, wu], 0)).float()) < 4e-3


def test_gemm_strided_views_and_alias(dev):
    from phantom_vlb_amd import ops
    M, N, K = 512, 256, 128
    big = _r(M, 3 * K, dev=dev)
    a = big[:, K:2 * K]                      # row stride 3K
    w = _r(N, K, dev=dev, scale=0.1)
    x = _r(M, N, dev=dev)
    ref = a.float() @ w.float().t() + x.float()
    ops.gemm(a, w, residual=x, out=x)        # residual aliases out
    assert rel_err(x, ref) < 6e-3


def test_gemm_rejects_bad_args(dev):
    from phantom_vlb_amd import ops
    from phantom_vlb_amd._lib import VlbError
    a, w = _r(8, 12, dev=dev), _r(8, 12, dev=dev)
    with pytest.raises(VlbError):
        ops.gemm(a, w)                       # K % 8 != 0


def test_transpose(dev):
    from phantom_vlb_amd import ops
    x = _r(300, 520, dev=dev)
    assert torch.equal(ops.transpose(x), x.t().contiguous())


# ------------------------------------------------------------------ attention
def _attn_ref(q, k, v, causal, key_mask, scale):
    B, S, Hq, D = q.shape
    rep = Hq // k.shape[2]
    qf, kf, vf = (t.float().transpose(1, 2) for t in (q, k, v))
    kf, vf = kf.repeat_interleave(rep, 1), vf.repeat_interleave(rep, 1)
    s = qf @ kf.transpose(2, 3) * scale
    allow = torch.ones(B, 1, S, S, dtype=torch.bool)
    if causal:
        allow = allow & torch.ones(S, S, dtype=torch.bool).tril()
    if key_mask is not None:
        allow = allow & key_mask.bool()[:, None, None, :]
    s = s.masked_fill(~allow, float("-inf"))
    return (torch.softmax(s, -1) @ vf).transpose(1, 2)   # [B,S,Hq,D]


@pytest.mark.parametrize("B,S,Hq,Hkv,D,causal,masked", [
    (2, 128, 4, 1, 128, True, True),
    (2, 320, 8, 2, 128, True, True),
    (1, 2048, 4, 1, 128, True, False),
    (3, 37, 2, 2, 64, False, False),
    (2, 577, 4, 4, 64, False, False),
    (1, 200, 2, 2, 64, True, True),
])
def test_attention_fwd(dev, B, S, Hq, Hkv, D, causal, masked):
    from phantom_vlb_amd import ops
    g = torch.Generator().manual_seed(S + D)
    qkv = torch.randn(B * S, (Hq + 2 * Hkv) * D, generator=g).to(BF)
    mask = None
    if masked:
        mask = torch.ones(B, S, dtype=torch.uint8)
        mask[0, S - S // 3:] = 0
    qd, kd = Hq * D, Hkv * D
    dq = qkv.to(dev)
    out, lse = ops.attention_fwd(dq[:, :qd], dq[:, qd:qd + kd], dq[:, qd + kd:], B, S, Hq, Hkv, D, causal, D ** -0.5,
                                 key_mask=None if mask is None else mask.to(dev), need_lse=True)
    q = qkv[:, :qd].view(B, S, Hq, D)
    k = qkv[:, qd:qd + kd].view(B, S, Hkv, D)
    v = qkv[:, qd + kd:].view(B, S, Hkv, D)
    ref = _attn_ref(q, k, v, causal, mask, D ** -0.5)
    got = out.view(B, S, Hq, D).float().cpu()
    valid = torch.ones(B, S, dtype=torch.bool)
    if masked and not causal:
        valid = mask.bool()
    # rows whose keys are all masked cannot occur with causal+right padding (key 0 is always valid)
    err = ((got - ref).abs() * valid[..., None, None]).max() / ref.abs().max()
    assert err < 1e-2, err
    assert torch.isfinite(got).all()
    # lse against the reference definition
    qf, kf = q.float().transpose(1, 2), k.float().transpose(1, 2).repeat_interleave(Hq // Hkv, 1)
    s = qf @ kf.transpose(2, 3) * D ** -0.5
    allow = torch.ones(B, 1, S, S, dtype=torch.bool)
    if causal:
        allow = allow & torch.ones(S, S, dtype=torch.bool).tril()
    if mask is not None:
        allow = allow & mask.bool()[:, None, None, :]
    lse_ref = torch.logsumexp(s.masked_fill(~allow, float("-inf")), -1)
    assert (lse.cpu() - lse_ref).abs().max() < 2e-2


def test_attention_softmax_rescale_branch(dev):
    """Spike one key against one query late in the sequence so the running max jumps mid-sweep."""
    from phantom_vlb_amd import ops
    B, S, H, D = 1, 256, 1, 128
    g = torch.Generator().manual_seed(5)
    q = torch.randn(B * S, D, generator=g) * 0.5
    k = torch.randn(B * S, D, generator=g) * 0.5
    v = torch.randn(B * S, D, generator=g)
    k[200] = q[230] * 6.0          # key 200 dominates query 230 (tile 3 of 4)
    q, k, v = (t.to(BF) for t in (q, k, v))
    out = ops.attention_fwd(q.to(dev), k.to(dev), v.to(dev), B, S, H, H, D, True, D ** -0.5)
    ref = _attn_ref(q.view(B, S, H, D), k.view(B, S, H, D), v.view(B, S, H, D), True, None, D ** -0.5)
    assert rel_err(out.view(B, S, H, D), ref) < 1e-2


# ------------------------------------------------------------------ norms & element-wise
# dims walk every kernel path: register-cached rows (<= 512, <= 1024 for LayerNorm, <= 4096) and the generic loop
@pytest.mark.parametrize("dim", [128, 512, 1024, 4096, 6144])
def test_rmsnorm(dev, dim):
    from phantom_vlb_amd import ops
    x, w = _r(37, dim, dev=dev, scale=2.0), (1 + 0.1 * torch.randn(dim)).to(BF).to(dev)
    y = ops.rmsnorm(x, w, 1e-5)
    xf = x.float()
    ref = w.float() * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5))
    assert rel_err(y, ref) < 8e-3


@pytest.mark.parametrize("dim", [512, 4096])
def test_rmsnorm_bwd(dev, dim):
    from phantom_vlb_amd import ops
    x, w, dy = _r(33, dim, dev=dev), (1 + 0.1 * torch.randn(dim)).to(BF).to(dev), _r(33, dim, dev=dev, seed=3)
    xin = _r(33, dim, dev=dev, seed=9)
    xr = x.float().cpu().requires_grad_(True)
    (w.float().cpu() * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5)) * dy.float().cpu()).sum().backward()
    dx = ops.rmsnorm_bwd(x, w, dy, 1e-5, dx_in=xin)
    assert rel_err(dx, xr.grad + xin.float().cpu()) < 8e-3


@pytest.mark.parametrize("act,res,dim", [(0, False, 128), (3, True, 128), (0, False, 1024), (3, True, 4096), (1, True, 5120)])
def test_layernorm(dev, act, res, dim):
    from phantom_vlb_amd import ops
    x = _r(50, dim, dev=dev, scale=3.0)
    w, b = (1 + 0.1 * torch.randn(dim)).to(BF).to(dev), (0.1 * torch.randn(dim)).to(BF).to(dev)
    r = _r(50, dim, dev=dev, seed=4) if res else None
    y = ops.layernorm(x, w, b, 1e-6, residual=r, act=act)
    ref = F.layer_norm(x.float(), (dim,), w.float(), b.float(), 1e-6)
    if res:
        ref = ref + r.float()
    if act == 3:
        ref = F.silu(ref)
    if act == 1:
        ref = ref * torch.sigmoid(1.702 * ref)
    assert rel_err(y, ref) < 8e-3


def test_rope(dev):
    import vlb_oracle as O
    from phantom_vlb_amd import ops
    B, S, H, D = 2, 40, 3, 128
    g = O.Geometry(head_dim=D, rope_theta=1e6)
    cos, sin = O.rope_tables(g, S)
    x = _r(B * S, H * D + 64, dev=dev)
    xr = x[:, :H * D].float().cpu().view(B, S, H, D).transpose(1, 2)
    ref = (xr * cos + O._rot_half(xr) * sin).transpose(1, 2).reshape(B * S, H * D)
    tail = x[:, H * D:].clone()
    ops.rope_(x, cos[:, :D // 2].contiguous().to(dev), sin[:, :D // 2].contiguous().to(dev), B, S, H, D)
    assert rel_err(x[:, :H * D], ref) < 8e-3
    assert torch.equal(x[:, H * D:], tail)
    # inverse rotation restores the input (size-independent property used by the backward)
    y = x.clone()
    ops.rope_(y, cos[:, :D // 2].contiguous().to(dev), sin[:, :D // 2].contiguous().to(dev), B, S, H, D, sign=-1)
    assert rel_err(y[:, :H * D], xr.transpose(1, 2).reshape(B * S, H * D)) < 1.6e-2


def test_swiglu_fwd_bwd(dev):
    from phantom_vlb_amd import ops
    gu, d = _r(70, 512, dev=dev), _r(70, 256, dev=dev, seed=2)
    out = ops.swiglu(gu)
    gr = gu.float().cpu().requires_grad_(True)
    ref = F.silu(gr[:, :256]) * gr[:, 256:]
    assert rel_err(out, ref) < 8e-3
    (ref * d.float().cpu()).sum().backward()
    assert rel_err(ops.swiglu_bwd(gu, d), gr.grad) < 8e-3


# ------------------------------------------------------------------ vision ingest / connector pieces
def test_patchify_is_conv_im2col(dev):
    from phantom_vlb_amd import ops
    N, H, P, Dv = 3, 84, 14, 64
    vis = torch.randn(N, 3, H, H)
    w = torch.randn(Dv, 3, P, P) * 0.05
    K, Kp = 3 * P * P, 640
    patches = ops.patchify(vis.to(dev), P, Kp)
    assert patches.shape == (N * 36, Kp)
    assert (patches[:, K:] == 0).all()
    wp = torch.zeros(Dv, Kp)
    wp[:, :K] = w.reshape(Dv, -1)
    got = ops.gemm(patches, wp.to(BF).to(dev))
    ref = F.conv2d(vis.to(BF).float(), w.to(BF).float(), stride=P).flatten(2).transpose(1, 2).reshape(N * 36, Dv)
    assert rel_err(got, ref) < 8e-3


def test_vit_assemble_and_drop_cls(dev):
    from phantom_vlb_amd import ops
    N, G, D = 3, 36, 128
    pe, cls, pos = _r(N * G, D, dev=dev), _r(D, dev=dev), _r(G + 1, D, dev=dev)
    tok = ops.vit_assemble(pe, cls, pos, N, G, D).view(N, G + 1, D)
    ref = torch.cat([cls.float().expand(N, 1, D), pe.float().view(N, G, D)], 1) + pos.float()[None]
    assert rel_err(tok, ref) < 5e-3
    assert torch.equal(ops.drop_cls(tok.view(-1, D), N, G, D).view(N, G, D), tok[:, 1:])


def test_dwconv_se_im2col(dev):
    from phantom_vlb_amd import ops
    N, H, C = 3, 13, 96
    x = _r(N * H * H, C, dev=dev)
    w = torch.randn(C, 1, 3, 3) * 0.3
    y = ops.dwconv3x3(x, w.flatten(1).t().contiguous().to(BF).to(dev), N, H, H, C)
    xn = x.float().cpu().view(N, H, H, C).permute(0, 3, 1, 2)
    ref = F.conv2d(xn, w.to(BF).float(), padding=1, groups=C).permute(0, 2, 3, 1).reshape(N * H * H, C)
    assert rel_err(y, ref) < 8e-3
    pooled = ops.se_pool(x, N, H * H, C)
    assert rel_err(pooled, x.float().view(N, H * H, C).mean(1)) < 8e-3
    gate = _r(N, C, dev=dev, seed=7)
    ys = ops.se_scale(x, gate, N, H * H, C)
    assert rel_err(ys, (x.float().view(N, H * H, C) * torch.sigmoid(gate.float())[:, None]).view(-1, C)) < 8e-3


def test_im2col3d_matches_conv3d(dev):
    from phantom_vlb_amd import ops
    B, T, H, C, Co = 2, 8, 6, 32, 48
    x = _r(B * T * H * H, C, dev=dev)
    w = torch.randn(Co, C, 2, 2, 2) * 0.1
    cols = ops.im2col3d(x, B, T, H, H, C)
    T2, H2 = T // 2 + 1, H // 2 + 1
    assert cols.shape == (B * T2 * H2 * H2, 8 * C)
    wk = w.permute(0, 2, 3, 4, 1).reshape(Co, -1).to(BF).to(dev)
    got = ops.gemm(cols, wk)
    xn = x.float().cpu().view(B, T, H, H, C).permute(0, 4, 1, 2, 3)
    ref = F.conv3d(xn, w.to(BF).float(), stride=2, padding=1).permute(0, 2, 3, 4, 1).reshape(-1, Co)
    assert rel_err(got, ref) < 8e-3


# ------------------------------------------------------------------ splice / weight mask vs the oracle
def test_splice_matches_oracle(dev):
    import vlb_oracle as O
    from phantom_vlb_amd import ops
    g = O.geometry_mini()
    batch = O.synthetic_batch(g, 4, seed=7)
    emb = (torch.randn(g.vocab, g.dim) * 0.02).to(BF)
    vid = torch.randn(4, g.vis_tokens, g.dim).to(BF)
    ids = batch["language"].long()
    ref, ref_mask = O.splice_multimodal(emb.float(), ids, vid.float())
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    out, mask = ops.splice_embed(ids.to(dev), emb.to(dev), vid.to(dev).view(-1, g.dim), g.vis_tokens, O.VIDEO_TOKEN_ID, err)
    assert int(err.item()) == 0
    assert torch.equal(out.view(4, g.max_len, g.dim).float().cpu(), ref)
    assert torch.equal(mask.bool().cpu(), ref_mask)
    # a row without its <video> token is reported, not silently accepted
    bad = ids.clone()
    bad[1][bad[1] == O.VIDEO_TOKEN_ID] = 5
    ops.splice_embed(bad.to(dev), emb.to(dev), vid.to(dev).view(-1, g.dim), g.vis_tokens, O.VIDEO_TOKEN_ID, err)
    assert int(err.item()) != 0


def test_weight_mask_matches_oracle_and_kat(dev):
    import numpy as np
    import os
    import vlb_oracle as O
    from phantom_vlb_amd import ops
    g = O.geometry_mini()
    batch = O.synthetic_batch(g, 4, seed=3)
    ref = O.make_weight_mask(batch["padvals"], batch["vis_weights"], batch["lang_weights"], g.lang_len, g.max_len,
                             g.ds_grid ** 2)
    got = ops.weight_mask(batch["padvals"].to(dev), batch["vis_weights"].to(dev), batch["lang_weights"].to(dev),
                          g.ds_grid ** 2, g.max_len, round_bf16=False)
    assert torch.equal(got.cpu(), ref)
    got_bf = ops.weight_mask(batch["padvals"].to(dev), batch["vis_weights"].to(dev), batch["lang_weights"].to(dev),
                             g.ds_grid ** 2, g.max_len, round_bf16=True)
    assert torch.equal(got_bf.cpu(), ref.to(BF).float())      # what the reference builds (bf16 mask)
    kat = np.load(os.path.join(os.path.dirname(__file__), "golden", "weight_mask_kat.npz"))
    for i in range(4):
        row = ops.weight_mask(torch.from_numpy(kat[f"{i}_padvals"]).to(dev), torch.from_numpy(kat[f"{i}_vis"]).to(dev),
                              torch.from_numpy(kat[f"{i}_lang"]).to(dev), 2, 20, round_bf16=False)
        assert torch.allclose(row.cpu()[0], torch.from_numpy(kat[f"{i}_row"]))


@pytest.mark.parametrize("tag", ["g7b", "g2f"])
def test_weight_mask_kernel_equals_the_reference_generated_rows(dev, tag):
    """tests/golden/ref_weight_mask.npz was written by the reference's OWN make_weight_mask
    (src/litmodule/videollama2_vlb_litmodule.py:178-203, run by oracle/gen_ref_fixtures.py): 40 (pad_len, inst_len,
    dialog_len) triples at the 7B geometry incl. (0,9,0), (300,9,58), (0,0,0), and a 2-frame geometry.  Bit-exact in bf16."""
    import numpy as np
    import os
    from phantom_vlb_amd import ops
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_weight_mask.npz"))
    pv, vw, lw = (torch.from_numpy(z[f"{tag}_{k}"]).to(dev) for k in ("padvals", "vis_weights", "lang_weights"))
    got = ops.weight_mask(pv, vw, lw, 169, int(z[f"{tag}_max_len"]), round_bf16=True)
    bits = got.cpu().to(BF).view(torch.int16).numpy().view(np.uint16)
    assert bits.shape == z[f"{tag}_rows_bf16_bits"].shape and np.array_equal(bits, z[f"{tag}_rows_bf16_bits"])
    assert torch.equal(got.cpu(), got.cpu().to(BF).float())        # round_bf16 rows are bf16-representable


# ------------------------------------------------------------------ brain head fwd + bwd vs oracle autograd
@pytest.mark.parametrize("B,S,E,V,drop", [(4, 128, 512, 128, False), (3, 96, 1024, 200, True), (5, 64, 4096, 256, False)])
def test_head_fwd_bwd(dev, B, S, E, V, drop):
    import vlb_oracle as O
    from phantom_vlb_amd.head import BrainHead, HEAD_PARAMS
    g = O.Geometry(dim=E, num_target=V, l2_lambda=1e-3)
    gen = torch.Generator().manual_seed(11)
    p = {"layer_norm1.weight": 1 + 0.1 * torch.randn(E, generator=gen), "layer_norm1.bias": 0.1 * torch.randn(E, generator=gen),
         "layer_norm2.weight": 1 + 0.1 * torch.randn(E, generator=gen), "layer_norm2.bias": 0.1 * torch.randn(E, generator=gen),
         "ridge_layer.linear.weight": torch.randn(V, E, generator=gen) / math.sqrt(E),
         "ridge_layer.linear.bias": 0.1 * torch.randn(V, generator=gen)}
    p = {k: v.to(BF).float() for k, v in p.items()}
    hidden = (torch.randn(B, S, E, generator=gen) * 2).to(BF)
    wm = torch.rand(B, S, generator=gen) * 0.1
    wm[:, : S // 3] = 0                         # skipped tokens
    wm[0, S // 2] = 0
    y = torch.randn(B, V, generator=gen)
    keep = None
    if drop:
        keep = (torch.rand(B, E, generator=gen) > 0.1).float() / 0.9
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    hr = hidden.float().requires_grad_(True)
    pred_ref, l2_ref, _ = O.brain_head(pr, hr, wm, g, keep_mask=None if keep is None else (keep > 0).float(), dropout_p=0.1 if drop else 0.0)
    loss_ref = F.mse_loss(pred_ref, y) + l2_ref
    loss_ref.backward()

    head = BrainHead(E, V, 1e-3, 1e-5, dev, sd=p)
    pred, terms = head.forward(hidden.to(dev).view(B * S, E), wm.to(dev), y.to(dev), None if keep is None else keep.to(dev))
    assert rel_err(pred, pred_ref) < 6e-3           # z is rounded to bf16 before the ridge GEMV (autocast semantics)
    t = terms.cpu()
    assert abs(float(t[1]) - float(l2_ref)) / float(l2_ref) < 1e-5
    assert abs(float(t[2]) - float(loss_ref)) / float(loss_ref) < 2e-3
    dh = head.backward(need_dhidden=True)
    for n in HEAD_PARAMS:
        assert rel_err(head.grads[n], pr[n].grad) < 1e-2, n
    assert rel_err(dh.view(B, S, E), hr.grad) < 1.5e-2
    assert (dh.view(B, S, E)[:, : S // 3] == 0).all()


@pytest.mark.parametrize("M,N,K,K2,plan", [(5015, 4096, 4096, 0, 256204), (2573, 6144, 4096, 0, 192203),
                                            (2600, 6144, 4096, 64, 192203), (5861, 4096, 14336, 64, 256202)])
def test_gemm_split_k_tail(dev, M, N, K, K2, plan):
    """Partial last wave cut along K through the workspace (vlb_gemm_bf16_ws): same result as the unsplit kernel up
    to fp32 summation order, deterministic, epilogue (bias / activation / residual / second pair) intact."""
    from phantom_vlb_amd import ops
    from phantom_vlb_amd._lib import lib
    assert lib.vlb_gemm_plan(M, N, K, K2, 1) == plan
    a, w = _r(M, K, dev=dev), _r(N, K, dev=dev, scale=0.05)
    bias, res = _r(N, dev=dev), _r(M, N, dev=dev)
    a2 = w2 = None
    ref = a.float() @ w.float().t()
    if K2:
        a2, w2 = _r(M, K2, dev=dev, seed=3), _r(N, K2, dev=dev, scale=0.05, seed=4)
        ref = ref + a2.float() @ w2.float().t()
    y = F.silu(ref + bias.float()) + res.float()
    assert ops.split_k_tails
    out = ops.gemm(a, w, bias=bias, residual=res, act=ops.ACT_SILU, a2=a2, w2=w2)
    assert rel_err(out, y) < 6e-3
    again = ops.gemm(a, w, bias=bias, residual=res, act=ops.ACT_SILU, a2=a2, w2=w2)
    assert torch.equal(out, again)
    ops.split_k_tails = False
    try:
        plain = ops.gemm(a, w, bias=bias, residual=res, act=ops.ACT_SILU, a2=a2, w2=w2)
    finally:
        ops.split_k_tails = True
    assert rel_err(plain, y) < 6e-3
    assert not torch.equal(plain, out)              # the tail really went through the split path
    assert rel_err(out, plain.float()) < 4e-3


def test_gemm_split_k_tail_swiglu_pair(dev):
    from phantom_vlb_amd import ops
    from phantom_vlb_amd._lib import lib
    M, ff, K = 5015, 2048, 4096
    assert lib.vlb_gemm_plan(M, 2 * ff, K, 0, 1) == 256204
    a = _r(M, K, dev=dev)
    wg, wu = _r(ff, K, dev=dev, scale=0.03, seed=1), _r(ff, K, dev=dev, scale=0.03, seed=2)
    out = ops.gemm(a, ops.interleave_gate_up(wg, wu), act=ops.ACT_SWIGLU_PAIR)
    ref = F.silu(a.float() @ wg.float().t()) * (a.float() @ wu.float().t())
    assert rel_err(out, ref) < 8e-3


def test_head_dropout_keep_scale_statistics_and_determinism(dev):
    """vlb_dropout_keep_scale: values are exactly 0 or 1/(1-p), keep rate p within binomial noise, a pure function
    of (seed, position), different seeds decorrelated."""
    from phantom_vlb_amd import ops
    B, E, p = 64, 4096, 0.1
    a = ops.dropout_keep_scale(B, E, p, 1234, dev)
    b = ops.dropout_keep_scale(B, E, p, 1234, dev)
    c = ops.dropout_keep_scale(B, E, p, 1235, dev)
    assert torch.equal(a, b) and not torch.equal(a, c)
    vals = torch.unique(a)
    assert vals.numel() == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1 / (1 - p)) < 1e-6
    keep = (a > 0).float()
    assert abs(float(keep.mean()) - (1 - p)) < 2e-3            # sigma = sqrt(.09/262144) = 6e-4
    agree = float(((a > 0) == (c > 0)).float().mean())
    assert abs(agree - (0.81 + 0.01)) < 5e-3                   # independent masks agree with prob p^2 + (1-p)^2
    rows = keep.mean(1)
    assert float(rows.std()) < 3 * (0.09 / E) ** 0.5 + 1e-3
    odd = ops.dropout_keep_scale(3, 5, 0.5, 7, dev)            # odd element count: the tail element is written
    assert odd.shape == (3, 5) and torch.isfinite(odd).all() and set(torch.unique(odd).tolist()) <= {0.0, 2.0}
