"""LoRA path: adapter kernels, attention backward, and the mini LoRA training step vs goldens (-m gpu)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
GOLD = os.path.join(os.path.dirname(__file__), "golden")


# ---- numpy restatement of the kernels' counter-based dropout mask (phantom_vlb_amd/csrc/lora.hip)
def _lowbias32(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & np.uint64(0xffffffff)
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & np.uint64(0xffffffff)
    x ^= x >> np.uint64(16)
    return x


def keep_mask(seed, M, K, p):
    thresh = min(65535, int(p * 65536 + 0.5))
    m = np.arange(M, dtype=np.uint64)[:, None]
    kp = np.arange(K // 2, dtype=np.uint64)[None, :]
    c = m * np.uint64(K // 2) + kp
    h = _lowbias32((c & np.uint64(0xffffffff)) ^ _lowbias32(((c >> np.uint64(32)) + np.uint64(seed)) & np.uint64(0xffffffff)))
    keep = np.empty((M, K), dtype=bool)
    keep[:, 0::2] = (h & np.uint64(0xffff)) >= thresh
    keep[:, 1::2] = (h >> np.uint64(16)) >= thresh
    return torch.from_numpy(keep)


def _r(*shape, dev, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(BF).to(dev)


@pytest.mark.parametrize("G,p", [(1, 0.0), (3, 0.0), (2, 0.1), (3, 0.25)])
def test_lora_down(dev, G, p):
    from phantom_vlb_amd.lora import lora_down
    M, K, R = 200, 256, 16 * G
    x, A = _r(M, K, dev=dev), _r(R, K, dev=dev, scale=0.1)
    seeds = [11 + 7 * g for g in range(G)]
    out = torch.zeros(M, 64, dtype=BF, device=dev)
    lora_down(x, A, R, 2.0, p, seeds, out)
    ref = torch.zeros(M, R)
    for g in range(G):
        xm = x.float().cpu()
        if p > 0:
            xm = xm * keep_mask(seeds[g], M, K, p) / (1 - p)
        ref[:, 16 * g:16 * g + 16] = 2.0 * xm @ A[16 * g:16 * g + 16].float().cpu().t()
    assert rel_err(out[:, :R], ref) < 8e-3
    assert (out[:, R:] == 0).all()


def test_dropout_mask_statistics(dev):
    """keep-rate of the counter-based mask, measured through the kernel itself (x = 1, A = 1)."""
    from phantom_vlb_amd.lora import lora_down
    M, K, p = 4096, 4096, 0.1
    x = torch.ones(M, K, dtype=BF, device=dev)
    A = torch.zeros(16, K, dtype=BF, device=dev)
    A[0] = 1
    out = torch.zeros(M, 64, dtype=BF, device=dev)
    lora_down(x, A, 16, 1.0, p, [12345], out)
    kept_per_row = out[:, 0].float().cpu() * (1 - p)                     # = number of kept columns
    rate = float(kept_per_row.sum()) / (M * K)
    assert abs(rate - 0.9) < 2e-3
    # rows are independent: the per-row keep count has binomial spread
    std = float(kept_per_row.std())
    assert 0.7 * math.sqrt(K * 0.09) < std < 1.5 * math.sqrt(K * 0.09) + 16   # bf16 rounding of the count adds noise
    # a different seed gives a different mask
    out2 = torch.zeros(M, 64, dtype=BF, device=dev)
    lora_down(x, A, 16, 1.0, p, [54321], out2)
    assert (out2[:, 0] != out[:, 0]).float().mean() > 0.5


@pytest.mark.parametrize("G,p", [(1, 0.1), (3, 0.1), (2, 0.0)])
def test_lora_dx_masked(dev, G, p):
    from phantom_vlb_amd.lora import lora_dx_masked
    M, K, R = 100, 1536, 16 * G
    u = torch.zeros(M, 64, dtype=BF, device=dev)
    u[:, :R] = _r(M, R, dev=dev)
    A = _r(R, K, dev=dev, scale=0.1)
    At = torch.zeros(K, 64, dtype=BF, device=dev)
    At[:, :R] = A.t()
    dx0 = _r(M, K, dev=dev, seed=5)
    dx = dx0.clone()
    seeds = [3 + g for g in range(G)]
    lora_dx_masked(u, At, dx, R, p, seeds)
    ref = dx0.float().cpu()
    for g in range(G):
        term = u[:, 16 * g:16 * g + 16].float().cpu() @ A[16 * g:16 * g + 16].float().cpu()
        if p > 0:
            term = term * keep_mask(seeds[g], M, K, p) / (1 - p)
        ref = ref + term
    assert rel_err(dx, ref) < 1e-2


@pytest.mark.parametrize("M,N,K,p", [(300, 16, 512, 0.0), (1000, 16, 1024, 0.1), (64, 48, 4096, 0.0), (700, 32, 776, 0.1),
                                     (2048, 48, 1024, 0.1)])
def test_wgrad_skinny(dev, M, N, K, p):
    from phantom_vlb_amd._lib import lib
    from phantom_vlb_amd.lora import wgrad_skinny
    Gm = torch.zeros(M, 64, dtype=BF, device=dev)
    Gm[:, :N] = _r(M, N, dev=dev)
    X = _r(M, K + 64, dev=dev, seed=2)[:, :K]                 # strided view
    dW = torch.full((N, K), 7.0, dtype=torch.float32, device=dev)
    ws = torch.empty(lib.vlb_wgrad_splits(M) * N * K, dtype=torch.float32, device=dev)
    seeds = [99 + 5 * g for g in range(N // 16)]
    wgrad_skinny(Gm, X, dW, ws, N, alpha=0.5, beta=0.0, p=p, seeds=seeds)
    ref = torch.zeros(N, K)
    for g in range(N // 16):
        xm = X.float().cpu()
        if p > 0:
            xm = xm * keep_mask(seeds[g], M, K, p) / (1 - p)
        ref[16 * g:16 * g + 16] = 0.5 * Gm[:, 16 * g:16 * g + 16].float().cpu().t() @ xm
    assert rel_err(dW, ref) < 2e-3
    wgrad_skinny(Gm, X, dW, ws, N, alpha=0.5, beta=1.0, p=p, seeds=seeds)     # accumulate
    assert rel_err(dW, 2 * ref) < 2e-3


@pytest.mark.parametrize("M,cols", [(300, [256]), (1000, [512, 256, 256]), (5861, [4096, 1024, 1024]), (2500, [1024, 1024])])
def test_wgrad_skinny_u_single_and_multi(dev, M, cols):
    """dB^T = t^T dY and u = s dY B in one sweep over dY: per projection (vlb_wgrad_skinny_u) and for the projections that share
    one dY in a single launch (vlb_wgrad_skinny_u_multi) - both against fp32 torch, and against each other."""
    from phantom_vlb_amd._lib import lib
    from phantom_vlb_amd.lora import wgrad_skinny_u, wgrad_skinny_u_multi
    n, K = len(cols), sum(cols)
    t = torch.zeros(M, 64, dtype=BF, device=dev)
    t[:, :16 * n] = _r(M, 16 * n, dev=dev)
    dy = _r(M, K + 64, dev=dev, seed=2)[:, :K]                # strided view, as dqkv's column slices are
    bts = [_r(16, c, dev=dev, scale=0.3, seed=10 + j) for j, c in enumerate(cols)]
    ws = torch.empty(lib.vlb_wgrad_splits(M) * 16 * K, dtype=torch.float32, device=dev)
    uws = torch.empty(lib.vlb_wgrad_u_ws_floats(M, K), dtype=torch.float32, device=dev)
    dws = [torch.full((16, c), 3.0, dtype=torch.float32, device=dev) for c in cols]
    u = torch.zeros(M, 64, dtype=BF, device=dev)
    wgrad_skinny_u_multi(t, dy, cols, dws, bts, ws, 2.0, u, uws)
    c0 = 0
    for j, c in enumerate(cols):
        dyj = dy[:, c0:c0 + c]
        ref_dw = t[:, 16 * j:16 * j + 16].float().t() @ dyj.float()
        ref_u = 2.0 * dyj.float() @ bts[j].float().t()
        assert rel_err(dws[j], ref_dw) < 2e-3, j
        assert rel_err(u[:, 16 * j:16 * j + 16], ref_u) < 1e-2, j
        dw1, u1 = torch.empty(16, c, dtype=torch.float32, device=dev), torch.zeros(M, 16, dtype=BF, device=dev)
        wgrad_skinny_u(t[:, 16 * j:16 * j + 16], dyj, dw1, ws, bts[j], 2.0, u1, uws)
        assert torch.equal(dw1, dws[j]) and torch.equal(u1, u[:, 16 * j:16 * j + 16]), j       # same sums in the same order
        c0 += c
    assert u[:, 16 * n:].abs().max() == 0


@pytest.mark.parametrize("B,S,Hq,Hkv,causal,masked", [(2, 128, 4, 1, True, True), (1, 300, 8, 2, True, False),
                                                     (2, 96, 2, 2, False, False), (1, 1024, 4, 1, True, True)])
def test_attention_bwd(dev, B, S, Hq, Hkv, causal, masked):
    from phantom_vlb_amd import ops
    from phantom_vlb_amd._lib import check, lib
    D = 128
    g = torch.Generator().manual_seed(S)
    qkv = (torch.randn(B * S, (Hq + 2 * Hkv) * D, generator=g) * 0.7).to(BF)
    dout = torch.randn(B * S, Hq * D, generator=g).to(BF)
    mask = None
    if masked:
        mask = torch.ones(B, S, dtype=torch.uint8)
        mask[0, S - S // 4:] = 0
    qd, kd = Hq * D, Hkv * D
    # reference: autograd through the fp32 definition
    q = qkv[:, :qd].float().view(B, S, Hq, D).requires_grad_(True)
    k = qkv[:, qd:qd + kd].float().view(B, S, Hkv, D).requires_grad_(True)
    v = qkv[:, qd + kd:].float().view(B, S, Hkv, D).requires_grad_(True)
    rep = Hq // Hkv
    s = q.transpose(1, 2) @ k.transpose(1, 2).repeat_interleave(rep, 1).transpose(2, 3) * D ** -0.5
    allow = torch.ones(B, 1, S, S, dtype=torch.bool)
    if causal:
        allow = allow & torch.ones(S, S, dtype=torch.bool).tril()
    if mask is not None:
        allow = allow & mask.bool()[:, None, None, :]
    o = (torch.softmax(s.masked_fill(~allow, float("-inf")), -1) @ v.transpose(1, 2).repeat_interleave(rep, 1)).transpose(1, 2)
    valid = torch.ones(B, S, 1, 1) if mask is None else mask.float()[:, :, None, None]
    (o * dout.float().view(B, S, Hq, D) * valid).sum().backward()      # padded query rows carry no gradient
    dq_ref, dk_ref, dv_ref = q.grad, k.grad, v.grad

    dq_ = qkv.to(dev)
    dmask = None if mask is None else mask.to(dev)
    dout_d = (dout.float().view(B, S, Hq, D) * valid).to(BF).view(B * S, Hq * D).to(dev)
    out, lse = ops.attention_fwd(dq_[:, :qd], dq_[:, qd:qd + kd], dq_[:, qd + kd:], B, S, Hq, Hkv, D, causal, D ** -0.5,
                                 key_mask=dmask, need_lse=True)
    dqkv = ops.attention_bwd(dq_, qd, kd, out, dout_d, lse, dmask, B, S, Hq, Hkv, D, causal, D ** -0.5)
    got = dqkv.float().cpu()
    assert rel_err(got[:, :qd].view(B, S, Hq, D), dq_ref) < 2e-2
    assert rel_err(got[:, qd:qd + kd].view(B, S, Hkv, D), dk_ref) < 2e-2
    assert rel_err(got[:, qd + kd:].view(B, S, Hkv, D), dv_ref) < 2e-2


def _lora_cfg(p=0.0):
    from phantom_vlb_amd.litmodule import VLBLitModuleConfig
    return VLBLitModuleConfig(model_path="none", freeze_backbone=False, use_lora=True, lora_r=16, lora_alpha=32,
                              lora_dropout=p, dropout_rate=0.0, num_target=128, l2_lambda=1e-3, lr=1e-4,
                              betas=[0.9, 0.999], eps=1e-8, weight_decay=1e-2, lr_scheduler_name="CosineAnnealingLR",
                              last_epoch=-1, t_max=50000, geometry="mini")


def test_mini_lora_training_step_vs_golden(dev):
    """configs[0] with LoRA (B != 0 so every gradient path is live), dropout off: loss, pred, head and
    adapter gradients against the committed goldens."""
    import vlb_oracle as O
    from phantom_vlb_amd.litmodule import VLBLitModule
    from phantom_vlb_amd.lora import LoraState
    g = O.geometry_mini()
    p = O.round_bf16(O.init_params(g, seed=1234, lora=True, lora_b_std=0.02))
    batch = O.synthetic_batch(g, 4, seed=1234)
    gold = np.load(os.path.join(GOLD, "mini_lora.npz"))
    m = VLBLitModule(_lora_cfg())
    m.configure_model(state_dict=p, head_state=p)
    m.lora = LoraState(m.geometry, m.backbone.w, 16, 32, 0.0, m.device, sd=p)
    m.configure_optimizers()
    loss = m.training_step(batch)
    assert abs(float(loss) - float(gold["loss"])) / float(gold["loss"]) < 1e-3
    assert rel_err(m.head.pred, torch.from_numpy(gold["pred"])) < 3e-2
    checked = 0
    for name in gold.files:
        if not name.startswith("grad::"):
            continue
        n = name[len("grad::"):]
        ref = torch.from_numpy(gold[name])
        if ".lora_" in n:
            got = m.lora.grads[n]
            if "lora_B" in n:
                got = got.t()
        else:
            got = m.head.grads[n]
        assert rel_err(got, ref) < 6e-2, n
        checked += 1
    assert checked >= 6 + 28
    # global gradient norm (what the clip sees)
    tot = math.sqrt(sum(float(t.double().pow(2).sum()) for t in list(m.head.grads.values()) + list(m.lora.grads.values())))
    assert abs(tot - float(gold["grad_global_norm"])) / float(gold["grad_global_norm"]) < 3e-2
    # optimiser step refreshes the derived adapter layouts
    opt = m.optimizer
    a_before = m.lora.layers[0]["qkv"]["At"].clone()
    opt.step()
    assert not torch.equal(a_before, m.lora.layers[0]["qkv"]["At"])
    assert torch.equal(m.lora.layers[0]["qkv"]["At"][:, :48], m.lora.layers[0]["qkv"]["A"].t())


def test_validation_step_applies_the_adapters(dev):
    """validation_step (reference :309-342 runs the peft model in eval mode: adapters on, dropout off).  With trained
    (B != 0) adapters and LoRA dropout configured, the validation loss is the golden adapted loss, not the loss of the
    bare frozen backbone, and it does not depend on the dropout seed."""
    import vlb_oracle as O
    from phantom_vlb_amd.litmodule import VLBLitModule
    from phantom_vlb_amd.lora import LoraState
    g = O.geometry_mini()
    p = O.round_bf16(O.init_params(g, seed=1234, lora=True, lora_b_std=0.02))
    batch = O.synthetic_batch(g, 4, seed=1234)
    gold = np.load(os.path.join(GOLD, "mini_lora.npz"))
    m = VLBLitModule(_lora_cfg())
    m.configure_model(state_dict=p, head_state=p)
    m.lora = LoraState(m.geometry, m.backbone.w, 16, 32, 0.1, m.device, sd=p)      # dropout configured: must be off in eval
    m.configure_optimizers()
    out = m.validation_step(batch)
    assert abs(float(out["loss"]) - float(gold["loss"])) / float(gold["loss"]) < 1e-3
    assert rel_err(out["brain_preds"], torch.from_numpy(gold["pred"])) < 3e-2
    again = m.validation_step(batch)
    assert torch.equal(out["brain_preds"], again["brain_preds"])
    # the bare backbone (adapters ignored) gives a measurably different prediction: this is what eval mode must NOT return
    lora, m.lora = m.lora, None
    bare = m.validation_step(batch)
    m.lora = lora
    assert rel_err(bare["brain_preds"], torch.from_numpy(gold["pred"])) > 2 * rel_err(out["brain_preds"], torch.from_numpy(gold["pred"]))
    assert m.training                                                                # mode restored


def test_mini_lora_dropout_matches_oracle(dev):
    """LoRA dropout 0.1: the oracle is fed the SAME counter-based masks (restated in numpy)."""
    import vlb_oracle as O
    from phantom_vlb_amd.litmodule import VLBLitModule
    from phantom_vlb_amd.lora import GROUPS, LoraState
    g = O.geometry_mini()
    p = O.round_bf16(O.init_params(g, seed=5, lora=True, lora_b_std=0.05))
    batch = O.synthetic_batch(g, 2, seed=6)
    m = VLBLitModule(_lora_cfg(0.1))
    m.configure_model(state_dict=p, head_state=p)
    m.lora = LoraState(m.geometry, m.backbone.w, 16, 32, 0.1, m.device, sd=p)
    m.configure_optimizers()
    loss = m.training_step(batch)
    # rebuild the masks the kernels used (step counter is 1 after the first forward)
    lens = m.backbone.row_layout(batch["language"], batch["padvals"]).lens   # the step ran on packed rows
    M = sum(lens)

    def dense(mask):                     # packed rows [M,K] -> [2,S,K]; padded rows never reach the loss
        out = torch.ones(2, g.max_len, mask.shape[1])
        out[0, :lens[0]], out[1, :lens[1]] = mask[:lens[0]], mask[lens[0]:]
        return out
    drop = {}
    for li in range(g.layers):
        idx = 0
        for gname, targets in GROUPS:
            for t in targets:
                seed = m.lora._seed(li, idx)
                K = m.lora.in_dims[t]
                drop[f"model.layers.{li}.{t}"] = dense(keep_mask(seed, M, K, 0.1).float() / 0.9)
                idx += 1
    names = O.trainable_names(p, False, True)
    pr = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in p.items()}
    loss_ref, _ = O.training_loss(pr, batch, g, lora_drop=drop)
    loss_ref.backward()
    assert abs(float(loss) - float(loss_ref)) / float(loss_ref) < 1e-3
    for n in ("model.layers.0.self_attn.q_proj.lora_A.weight", "model.layers.1.mlp.down_proj.lora_A.weight",
              "model.layers.0.mlp.up_proj.lora_B.weight", "model.layers.1.self_attn.v_proj.lora_B.weight"):
        got = m.lora.grads[n].t() if "lora_B" in n else m.lora.grads[n]
        assert rel_err(got, pr[n].grad) < 6e-2, n


@pytest.mark.parametrize("M,N,K", [
    (700, 512, 256),          # a handful of 256-row tiles
    (5861, 4096, 4096),       # the o-proj dgrad at the LoRA batch: 192-row tiles, 496 = two waves
    (3000, 1024, 4096),
    (4500, 4096, 256),        # 192-row tiles with a re-cut tail (384 = 256 + 128 -> 192x128 halves)
    (5861, 28672, 128),       # 256-row tiles with a re-cut tail (2576 = 10 x 256 + 16 -> 256x128 halves)
    (5009, 4096, 4096),       # 256-row tiles: the masked kernel keeps the re-cut halves here (no split-K at 256 rows)
    (2573, 6144, 4096),       # 192-row tiles, partial wave split 3 ways along K (the masked pair rides in split 0)
])
def test_gemm_masked_pair_equals_gemm_plus_lora_dx(dev, M, N, K):
    """dx = dy.W + keep*(u.A)/(1-p) in one GEMM (mask applied to the LoRA accumulators in place) against the
    two-kernel path; the mask bits must be the very same (forward used them), the values agree to bf16 rounding."""
    from phantom_vlb_amd import ops
    from phantom_vlb_amd.lora import PAD, lora_dx_masked
    g = torch.Generator().manual_seed(M + N)
    dy = (torch.randn(M, K, generator=g) * 0.5).to(BF).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(dev)
    u = torch.zeros(M, PAD, dtype=BF)
    u[:, :16] = (torch.randn(M, 16, generator=g) * 0.5).to(BF)
    At = torch.zeros(N, PAD, dtype=BF)
    At[:, :16] = (torch.randn(N, 16, generator=g) * 0.2).to(BF)
    u, At = u.to(dev), At.to(dev)
    seed, p = 0x1234ABCD, 0.1
    assert ops.gemm_masked_pair_ok(M, N, K)
    fused = ops.gemm_masked_pair(dy, W, u, At, p, seed)
    base = ops.gemm(dy, W)
    two = base.clone()
    lora_dx_masked(u, At, two, 16, p, [seed])
    # reference in fp32 with the mask recovered from the two-kernel path: delta != 0 <=> kept (u.A is dense)
    lora = (u[:, :16].float() @ At[:, :16].float().t()) / (1 - p)
    keep = keep_mask(seed, M, N, p).to(dev)
    ref = dy.float() @ W.float().t() + lora * keep
    assert rel_err(fused, ref) < 6e-3
    assert rel_err(two, ref) < 1.2e-2            # the two-kernel path rounds to bf16 twice
    kept_frac = float(keep.float().mean())
    assert abs(kept_frac - 0.9) < 0.01


@pytest.mark.parametrize("M,ff,K", [(700, 512, 256), (5861, 14336, 4096), (4500, 4096, 256), (2573, 6144, 4096)])
def test_gemm_masked_pair_swiglu_bwd_matches_two_kernels(dev, M, ff, K):
    """Down-projection dgrad with the SwiGLU backward in its epilogue == masked-pair GEMM followed by vlb_swiglu_bwd
    (which rounds d_h to bf16 in between), and == the fp32 formula."""
    from phantom_vlb_amd import ops
    from phantom_vlb_amd.lora import PAD
    g = torch.Generator().manual_seed(M + ff)
    dy = (torch.randn(M, K, generator=g) * 0.5).to(BF).to(dev)
    W = (torch.randn(ff, K, generator=g) * 0.05).to(BF).to(dev)
    gu = torch.randn(M, 2 * ff, generator=g).to(BF).to(dev)
    u = torch.zeros(M, PAD, dtype=BF)
    u[:, :16] = (torch.randn(M, 16, generator=g) * 0.5).to(BF)
    At = torch.zeros(ff, PAD, dtype=BF)
    At[:, :16] = (torch.randn(ff, 16, generator=g) * 0.2).to(BF)
    u, At = u.to(dev), At.to(dev)
    seed, p = 0xBEEF1234, 0.1
    fused = ops.gemm_masked_pair_swiglu_bwd(dy, W, gu, u, At, p, seed)
    d_h = ops.gemm_masked_pair(dy, W, u, At, p, seed)
    two = ops.swiglu_bwd(gu, d_h)
    assert rel_err(fused, two.float()) < 6e-3
    keep = keep_mask(seed, M, ff, p).to(dev)
    dh = dy.float() @ W.float().t() + (u[:, :16].float() @ At[:, :16].float().t()) / (1 - p) * keep
    gate, up = gu[:, :ff].float(), gu[:, ff:].float()
    sg = torch.sigmoid(gate)
    ref = torch.cat([dh * up * sg * (1 + gate * (1 - sg)), dh * gate * sg], dim=1)
    assert rel_err(fused, ref) < 6e-3
    again = ops.gemm_masked_pair_swiglu_bwd(dy, W, gu, u, At, p, seed)
    assert torch.equal(fused, again)


@pytest.mark.parametrize("r,p", [(40, 0.0), (32, 0.1), (64, 0.1), (8, 0.1)])
def test_mini_lora_other_ranks_match_oracle(dev, r, p):
    """The reference's lora_r is a free int (litmodule :141): ranks that need several 16-wide MFMA blocks per projection
    (32, 40 - last block padded -, 64) and a rank below one block (8), without and with dropout (the c blocks of a
    projection share one mask), every adapter gradient against oracle autograd fed the same masks."""
    import dataclasses
    import vlb_oracle as O
    from phantom_vlb_amd.litmodule import VLBLitModule
    from phantom_vlb_amd.lora import GROUPS
    g = O.geometry_mini(lora_r=r, lora_alpha=32)
    pp = O.round_bf16(O.init_params(g, seed=5, lora=True, lora_b_std=0.05))
    batch = O.synthetic_batch(g, 2, seed=6)
    cfg = dataclasses.replace(_lora_cfg(p), lora_r=r)
    m = VLBLitModule(cfg)
    m.configure_model(state_dict=pp)
    assert m.lora.c == (r + 15) // 16 and m.lora.layers[0]["qkv"]["R"] == 3 * 16 * m.lora.c
    opt, _ = m.configure_optimizers()
    loss = m.training_step(batch)
    lens = m.backbone.row_layout(batch["language"], batch["padvals"]).lens
    M = sum(lens)

    def dense(mask):
        out = torch.ones(2, g.max_len, mask.shape[1])
        out[0, :lens[0]], out[1, :lens[1]] = mask[:lens[0]], mask[lens[0]:]
        return out
    drop = None
    if p > 0:
        drop = {}
        for li in range(g.layers):
            idx = 0
            for gname, targets in GROUPS:
                for t in targets:
                    drop[f"model.layers.{li}.{t}"] = dense(keep_mask(m.lora._seed(li, idx), M, m.lora.in_dims[t], p).float() / (1 - p))
                    idx += 1
    names = O.trainable_names(pp, False, True)
    pr = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in pp.items()}
    loss_ref, _ = O.training_loss(pr, batch, g, lora_drop=drop)
    loss_ref.backward()
    assert abs(float(loss) - float(loss_ref)) / float(loss_ref) < 1e-3
    sd_grads = {n: (t[:r].t() if "lora_B" in n else t[:r]) for n, t in m.lora.grads.items()}
    worst = max(rel_err(sd_grads[n], pr[n].grad) for n in sd_grads)
    assert worst < 6e-2, worst
    for n, t in m.lora.grads.items():                         # padded rank rows receive exactly zero gradient
        if t.shape[0] > r:
            assert float(t[r:].abs().max()) == 0.0, n
    opt[0].step()
    sd = m.trainable_state_dict()
    assert sd["model.layers.0.self_attn.k_proj.lora_A.weight"].shape == (r, g.dim)
    assert sd["model.layers.0.self_attn.k_proj.lora_B.weight"].shape == (g.kv_heads * g.head_dim, r)
