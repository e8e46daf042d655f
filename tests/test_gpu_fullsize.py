"""Parity at BASELINE.json's FULL sizes (7B geometry: d=4096, 32/8 heads of 128, ff=14336, S=2048, V=2048).
Where the oracle finishes in seconds it is used directly (head, one decoder layer, splice, mask); the
rest is checked on sampled rows against fp32 torch, plus size-independent properties."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def test_head_full_size_vs_oracle(dev):
    """B=5, S=2048, E=4096, V=2048: loss, prediction, every head gradient and d hidden vs oracle autograd."""
    import vlb_oracle as O
    from phantom_vlb_amd.head import HEAD_PARAMS, BrainHead
    B, S, E, V = 5, 2048, 4096, 2048
    g = O.Geometry(dim=E, num_target=V)
    gen = torch.Generator().manual_seed(1)
    p = O.round_bf16({n: t for n, t in O.init_params(O.geometry_mini(dim=E, num_target=V), seed=2).items() if n in HEAD_PARAMS})
    hidden = (torch.randn(B, S, E, generator=gen) * 1.5).to(BF)
    batch = O.synthetic_batch(O.geometry_7b(), B, seed=3)
    wm = O.make_weight_mask(batch["padvals"], batch["vis_weights"], batch["lang_weights"], 866, 2048).to(BF).float()
    y = torch.randn(B, V, generator=gen)
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    hr = hidden.float().requires_grad_(True)
    pred_ref, l2_ref, _ = O.brain_head(pr, hr, wm, g)
    loss_ref = F.mse_loss(pred_ref, y) + l2_ref
    loss_ref.backward()
    head = BrainHead(E, V, g.l2_lambda, g.ln_eps, dev, sd=p)
    pred, terms = head.forward(hidden.to(dev).view(B * S, E), wm.to(dev), y.to(dev))
    assert abs(float(terms[2]) - float(loss_ref)) / float(loss_ref) < 1e-3
    assert rel_err(pred, pred_ref) < 8e-3
    dh = head.backward(need_dhidden=True)
    for n in HEAD_PARAMS:
        assert rel_err(head.grads[n], pr[n].grad) < 1.5e-2, n
    assert rel_err(dh.view(B, S, E), hr.grad) < 2e-2
    # tokens with zero HRF weight (prompt, instruction, padding) carry exactly zero gradient
    assert (dh.view(B, S, E)[wm.to(dev) == 0] == 0).all()


def test_head_whole_cortex_size_vs_oracle(dev):
    """The largest head BASELINE.json names (configs[4]: 65,536 targets; W is 537 MB in bf16): prediction, loss and
    every gradient vs oracle autograd at B=2."""
    import vlb_oracle as O
    from phantom_vlb_amd.head import HEAD_PARAMS, BrainHead
    B, S, E, V = 2, 2048, 4096, 65536
    g = O.Geometry(dim=E, num_target=V)
    gen = torch.Generator().manual_seed(11)
    p = O.round_bf16({n: t for n, t in O.init_params(O.geometry_mini(dim=E, num_target=V), seed=5).items() if n in HEAD_PARAMS})
    hidden = (torch.randn(B, S, E, generator=gen) * 1.5).to(BF)
    batch = O.synthetic_batch(O.geometry_7b(), B, seed=7)
    wm = O.make_weight_mask(batch["padvals"], batch["vis_weights"], batch["lang_weights"], 866, 2048).to(BF).float()
    y = torch.randn(B, V, generator=gen)
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    hr = hidden.float().requires_grad_(True)
    pred_ref, l2_ref, _ = O.brain_head(pr, hr, wm, g)
    loss_ref = F.mse_loss(pred_ref, y) + l2_ref
    loss_ref.backward()
    head = BrainHead(E, V, g.l2_lambda, g.ln_eps, dev, sd=p)
    pred, terms = head.forward(hidden.to(dev).view(B * S, E), wm.to(dev), y.to(dev))
    assert abs(float(terms[2]) - float(loss_ref)) / float(loss_ref) < 1e-3
    assert abs(float(terms[1]) - float(l2_ref)) / float(l2_ref) < 1e-3
    assert rel_err(pred, pred_ref) < 8e-3
    dh = head.backward(need_dhidden=True)
    for n in HEAD_PARAMS:
        assert rel_err(head.grads[n], pr[n].grad) < 1.5e-2, n
    assert rel_err(dh.view(B, S, E), hr.grad) < 2e-2
    del head
    torch.cuda.empty_cache()


def test_decoder_layer_full_size_vs_oracle(dev):
    """One Mistral-7B-sized decoder layer (+ final norm) on one 2048-token clip with 150 padded positions."""
    import dataclasses
    import vlb_oracle as O
    from phantom_vlb_amd.backbone import Backbone, Weights
    from phantom_vlb_amd.geometry import geometry_7b
    go = dataclasses.replace(O.geometry_7b(), layers=1, vit_layers=2, proj_depth=1)
    full = O.init_params(go, seed=5)
    p = O.round_bf16({k: v for k, v in full.items() if k.startswith(("model.layers.0.", "model.norm"))})
    S = 2048
    gen = torch.Generator().manual_seed(6)
    x = (torch.randn(1, S, go.dim, generator=gen) * 0.5).to(BF)
    mask = torch.ones(1, S, dtype=torch.bool)
    mask[0, S - 150:] = False
    with torch.no_grad():
        ref = O.mistral_decoder(p, x.float(), mask, go)
    # device side: build only the decoder part of the weights
    g = dataclasses.replace(geometry_7b(), layers=1)
    w = Weights.__new__(Weights)
    w.g, w.dev = g, dev
    lw = {"in_norm": p["model.layers.0.input_layernorm.weight"], "post_norm": p["model.layers.0.post_attention_layernorm.weight"],
          "wqkv": torch.cat([p[f"model.layers.0.self_attn.{n}_proj.weight"] for n in "qkv"], 0),
          "wo": p["model.layers.0.self_attn.o_proj.weight"], "wdown": p["model.layers.0.mlp.down_proj.weight"]}
    from phantom_vlb_amd import ops
    lw = {k: v.to(dev, BF).contiguous() for k, v in lw.items()}
    lw["wgu_il"] = ops.interleave_gate_up(p["model.layers.0.mlp.gate_proj.weight"].to(dev, BF),
                                           p["model.layers.0.mlp.up_proj.weight"].to(dev, BF))
    w.layers = [lw]
    w.final_norm = p["model.norm.weight"].to(dev, BF)
    inv = 1.0 / (g.rope_theta ** (torch.arange(0, g.head_dim, 2, dtype=torch.float32) / g.head_dim))
    fr = torch.arange(S, dtype=torch.float32)[:, None] * inv[None]
    w.rope_cos, w.rope_sin = fr.cos().to(dev).contiguous(), fr.sin().to(dev).contiguous()
    bb = Backbone(g, w)
    out = bb.decoder(x.to(dev).view(S, g.dim).clone(), mask.to(torch.uint8).to(dev), 1, S)
    got = out.view(1, S, g.dim).float().cpu()
    err = ((got - ref).abs() * mask[..., None]).max() / ref.abs().max()
    assert err < 2e-2, err
    assert torch.isfinite(got).all()


def test_vision_tower_and_connector_full_size_vs_oracle(dev):
    """One 12-frame clip through the ViT-L/14-336-wide tower (2 of its 23 layers: 577 tokens, 16 heads of 64,
    patch-embed K=588) and the FULL STC connector (2 x 4 RegStage blocks at 4096 channels, Conv3d 12x24x24 ->
    7x13x13, MLP readout) against the fp32 oracle: the frozen vision path at BASELINE's real widths."""
    import dataclasses
    import vlb_oracle as O
    from phantom_vlb_amd.backbone import Backbone, Weights
    from phantom_vlb_amd.geometry import geometry_7b
    go = dataclasses.replace(O.geometry_7b(), layers=1, vit_layers=3)
    p = O.round_bf16(O.init_params(go, seed=8))
    g = dataclasses.replace(geometry_7b(), layers=1, vit_layers=3)
    assert g.vit_layers_run == 2 and g.vis_tokens == 1183
    gen = torch.Generator().manual_seed(9)
    pix = torch.randn(g.num_frames, 3, g.image_size, g.image_size, generator=gen)
    with torch.no_grad():
        feats_ref = O.clip_tower(p, pix, go)
        vid_ref = O.stc_connector(p, feats_ref.view(1, g.num_frames, g.grid * g.grid, g.vit_dim), go)
    bb = Backbone(g, Weights(g, p, dev))
    feats = bb.vision_tower(pix.to(dev))
    assert feats.shape == (g.num_frames * g.grid * g.grid, g.vit_dim)
    assert rel_err(feats.view(g.num_frames, -1, g.vit_dim), feats_ref) < 2e-2
    vid = bb.connector(feats, 1)
    assert vid.shape == (g.vis_tokens, g.dim)
    assert rel_err(vid.view(1, g.vis_tokens, g.dim), vid_ref) < 5e-2
    assert torch.isfinite(vid.float()).all()
    del bb
    torch.cuda.empty_cache()


def test_gemm_full_size_sampled_rows(dev):
    """gate/up and down projections at M = 10240: 512 sampled rows against fp32 torch on the GPU."""
    from phantom_vlb_amd import ops
    gen = torch.Generator(device=dev).manual_seed(0)
    for (M, N, K) in ((10240, 28672, 4096), (10240, 4096, 14336), (34620, 3072, 1024)):
        a = torch.randn(M, K, device=dev, generator=gen).to(BF)
        w = (torch.randn(N, K, device=dev, generator=gen) * 0.02).to(BF)
        res = torch.randn(M, N, device=dev, generator=gen).to(BF)
        out = ops.gemm(a, w, residual=res)
        rows = torch.randint(0, M, (512,), device=dev, generator=gen)
        rows[0], rows[1] = 0, M - 1
        ref = a[rows].float() @ w.float().t() + res[rows].float()
        assert rel_err(out[rows], ref) < 6e-3
        # linearity in A (size-independent property): gemm(2a) - 2 gemm(a) == 0 up to bf16 rounding of outputs
        out2 = ops.gemm(a * 2, w)
        out1 = ops.gemm(a, w)
        assert rel_err(out2.float()[:2048], 2 * out1.float()[:2048]) < 8e-3


def test_attention_full_size(dev):
    """B=1, S=2048, 32 q-heads / 8 kv-heads of 128, causal + padded tail: all heads vs fp32 torch on the GPU."""
    from phantom_vlb_amd import ops
    B, S, Hq, Hkv, D = 1, 2048, 32, 8, 128
    gen = torch.Generator(device=dev).manual_seed(2)
    qkv = (torch.randn(B * S, (Hq + 2 * Hkv) * D, device=dev, generator=gen) * 0.6).to(BF)
    mask = torch.ones(B, S, dtype=torch.uint8, device=dev)
    mask[0, S - 211:] = 0
    qd, kd = Hq * D, Hkv * D
    out = ops.attention_fwd(qkv[:, :qd], qkv[:, qd:qd + kd], qkv[:, qd + kd:], B, S, Hq, Hkv, D, True, D ** -0.5, key_mask=mask)
    q = qkv[:, :qd].float().view(B, S, Hq, D).transpose(1, 2)
    k = qkv[:, qd:qd + kd].float().view(B, S, Hkv, D).transpose(1, 2).repeat_interleave(Hq // Hkv, 1)
    v = qkv[:, qd + kd:].float().view(B, S, Hkv, D).transpose(1, 2).repeat_interleave(Hq // Hkv, 1)
    allow = torch.ones(S, S, dtype=torch.bool, device=dev).tril() & mask.bool()[0][None, :]
    s = (q @ k.transpose(2, 3)) * D ** -0.5
    ref = (torch.softmax(s.masked_fill(~allow, float("-inf")), -1) @ v).transpose(1, 2)
    got = out.view(B, S, Hq, D).float()
    valid = mask.bool()[0]
    assert float((got - ref)[:, valid].abs().max() / ref.abs().max()) < 1e-2
    assert torch.isfinite(got).all()


def test_splice_and_mask_full_size_bit_exact(dev):
    import vlb_oracle as O
    from phantom_vlb_amd import ops
    g = O.geometry_7b()
    batch = O.synthetic_batch(dataclass_small_vision(g), 5, seed=11)
    emb = (torch.randn(g.vocab, 256) * 0.02).to(BF)            # narrow embedding: the gather logic is width-independent
    vid = torch.randn(5, g.vis_tokens, 256).to(BF)
    ids = batch["language"].long()
    ref, ref_mask = O.splice_multimodal(emb.float(), ids, vid.float())
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    out, mask = ops.splice_embed(ids.to(dev), emb.to(dev), vid.to(dev).view(-1, 256), g.vis_tokens, O.VIDEO_TOKEN_ID, err)
    assert int(err.item()) == 0 and out.shape == (5 * 2048, 256)
    assert torch.equal(out.view(5, 2048, 256).float().cpu(), ref) and torch.equal(mask.bool().cpu(), ref_mask)
    wm_ref = O.make_weight_mask(batch["padvals"], batch["vis_weights"], batch["lang_weights"], 866, 2048)
    wm = ops.weight_mask(batch["padvals"].to(dev), batch["vis_weights"].to(dev), batch["lang_weights"].to(dev), 169, 2048,
                         round_bf16=False)
    assert torch.equal(wm.cpu(), wm_ref)


def dataclass_small_vision(g):
    """7B token geometry but 14x14-pixel frames, so the synthetic batch does not allocate 80 MB of pixels."""
    import dataclasses

    class _G:
        pass
    small = dataclasses.replace(g, image_size=14)
    # lang_len / vis_tokens must stay those of the 7B geometry: override through a thin proxy
    proxy = _G()
    for f in ("num_frames", "vocab", "num_target"):
        setattr(proxy, f, getattr(g, f))
    proxy.image_size = 14
    proxy.lang_len, proxy.ds_frames = g.lang_len, g.ds_frames
    return proxy


def test_step_is_deterministic(dev):
    """Two identical steps from identical state give bit-identical loss and gradients (fixed-order reductions)."""
    from phantom_vlb_amd.litmodule import VLBLitModule, VLBLitModuleConfig
    from phantom_vlb_amd.synthetic import synthetic_batch
    cfg = VLBLitModuleConfig(model_path="none", freeze_backbone=True, use_lora=False, lora_r=None, lora_alpha=None,
                             lora_dropout=None, dropout_rate=0.0, num_target=128, l2_lambda=1e-3, lr=1e-3, betas=[0.9, 0.999],
                             eps=1e-8, weight_decay=1e-2, lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000,
                             geometry="mini")
    m = VLBLitModule(cfg)
    m.configure_model()
    m.configure_optimizers()
    batch = synthetic_batch(m.geometry, 4, seed=1, device=m.device)
    l1 = float(m.training_step(batch)); g1 = m.flat.grad.clone()
    l2 = float(m.training_step(batch)); g2 = m.flat.grad.clone()
    assert l1 == l2 and torch.equal(g1, g2)


def test_lora_step_is_deterministic(dev):
    """LoRA step (dropout on): forward, attention backward (dQ in its own register-accumulating pass, no
    atomics), skinny wgrads and the head are all fixed-order -> two runs from the same state are bit-identical."""
    from phantom_vlb_amd.litmodule import VLBLitModule, VLBLitModuleConfig
    from phantom_vlb_amd.synthetic import synthetic_batch
    cfg = VLBLitModuleConfig(model_path="none", freeze_backbone=False, use_lora=True, lora_r=16, lora_alpha=32,
                             lora_dropout=0.1, dropout_rate=0.0, num_target=128, l2_lambda=1e-3, lr=1e-3, betas=[0.9, 0.999],
                             eps=1e-8, weight_decay=1e-2, lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000,
                             geometry="mini")
    m = VLBLitModule(cfg)
    m.configure_model()
    m.configure_optimizers()
    batch = synthetic_batch(m.geometry, 3, seed=2)
    out = []
    for _ in range(2):
        m.lora.step = 0                           # same dropout seeds for both runs
        loss = float(m.training_step(batch))
        out.append((loss, m.flat.grad.clone()))
    assert out[0][0] == out[1][0]
    assert torch.equal(out[0][1], out[1][1])


def test_7b_step_packed_equals_dense(dev):
    """BASELINE configs[1] at full size (7B frozen backbone, 2048 targets, max_len 2048, B=2): the
    unpadded row layout gives the SAME loss and head gradients, bit for bit, as the dense layout."""
    from phantom_vlb_amd.litmodule import VLBLitModule, VLBLitModuleConfig
    from phantom_vlb_amd.synthetic import synthetic_batch
    cfg = VLBLitModuleConfig(model_path="none", freeze_backbone=True, use_lora=False, lora_r=None, lora_alpha=None,
                             lora_dropout=None, dropout_rate=0.0, num_target=2048, l2_lambda=1e-3, lr=1e-4,
                             betas=[0.9, 0.999], eps=1e-8, weight_decay=1e-2, lr_scheduler_name="CosineAnnealingLR",
                             last_epoch=-1, t_max=50000, geometry="7b")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = VLBLitModule(cfg)
        m.configure_model()
    m.configure_optimizers()
    batch = synthetic_batch(m.geometry, 2, seed=3, device=m.device)
    batch["language"], batch["padvals"] = batch["language"].cpu(), batch["padvals"].cpu()
    lay = m.backbone.row_layout(batch["language"], batch["padvals"])
    assert lay.rows < 2 * m.geometry.max_len            # the synthetic clips do carry padding
    from phantom_vlb_amd import ops
    res = {}
    ops.split_k_tails = False           # one K order for every tile: a row's result is independent of the row count
    try:
        for pack in (False, True):
            m.pack_tokens = pack
            loss = float(m.training_step(batch))
            res[pack] = (loss, m.flat.grad.clone(), m.head.pred.clone())
    finally:
        ops.split_k_tails = True
    assert res[True][0] == res[False][0], (res[True][0], res[False][0])
    assert torch.equal(res[True][2], res[False][2])
    assert torch.equal(res[True][1], res[False][1])
    # default configuration (partial waves of tiles split along K through the workspace): same step up to the fp32
    # summation order of those tiles
    m.pack_tokens = True
    loss_sk = float(m.training_step(batch))
    assert abs(loss_sk - res[True][0]) <= 2e-3 * abs(res[True][0]), (loss_sk, res[True][0])
    assert (m.head.pred.float() - res[True][2].float()).abs().max() <= 2e-2 * res[True][2].float().abs().max()
    del m
    torch.cuda.empty_cache()


@pytest.mark.parametrize("fused_mlp_in", [True, False])
def test_lora_decoder_layer_full_size_fwd_bwd_vs_oracle(dev, fused_mlp_in):
    """One Mistral-7B-sized decoder layer with LoRA r=16 (dropout 0.1) on one 2048-token clip, 150 padded positions:
    forward output and the gradients of all 14 adapter matrices against the oracle's autograd, the oracle being fed
    the very masks the kernels used.  Exercises at BASELINE sizes: fused LoRA GEMMs (second operand pair), the
    four-wave GEMM with 192/256-row tiles and re-cut tails, attention forward, dK/dV + GQA reduce + dQ, the fused
    dB/u skinny wgrad, the masked-pair dgrad GEMM and lora_dx."""
    import dataclasses
    import vlb_oracle as O
    from phantom_vlb_amd import ops
    from phantom_vlb_amd.backbone import Backbone, Weights
    from phantom_vlb_amd.geometry import geometry_7b
    from phantom_vlb_amd.lora import GROUPS, LoraState
    from test_gpu_lora import keep_mask
    go = dataclasses.replace(O.geometry_7b(lora_r=16, lora_alpha=32), layers=1, vit_layers=2, proj_depth=1)
    full = O.init_params(go, seed=11, lora=True, lora_b_std=0.02)
    p = O.round_bf16({k: v for k, v in full.items() if k.startswith(("model.layers.0.", "model.norm"))})
    S = 2048
    gen = torch.Generator().manual_seed(12)
    x = (torch.randn(1, S, go.dim, generator=gen) * 0.5).to(BF)
    gout = (torch.randn(1, S, go.dim, generator=gen) * 0.1).to(BF)
    mask = torch.ones(1, S, dtype=torch.bool)
    mask[0, S - 150:] = False
    gout[0, S - 150:] = 0                                  # the head never weights padded rows
    # ---- device: one-layer weights (with the transposed copies LoRA's dgrad needs) + adapters from the same dict
    g = dataclasses.replace(geometry_7b(lora_r=16, lora_alpha=32), layers=1)
    w = Weights.__new__(Weights)
    w.g, w.dev = g, dev
    lw = {"in_norm": p["model.layers.0.input_layernorm.weight"], "post_norm": p["model.layers.0.post_attention_layernorm.weight"],
          "wqkv": torch.cat([p[f"model.layers.0.self_attn.{n}_proj.weight"] for n in "qkv"], 0),
          "wo": p["model.layers.0.self_attn.o_proj.weight"], "wdown": p["model.layers.0.mlp.down_proj.weight"],
          "wgu": torch.cat([p["model.layers.0.mlp.gate_proj.weight"], p["model.layers.0.mlp.up_proj.weight"]], 0)}
    lw = {k: v.to(dev, BF).contiguous() for k, v in lw.items()}
    for k in ("wqkv", "wo", "wgu", "wdown"):
        lw[k + "_t"] = ops.transpose(lw[k])
    if fused_mlp_in:        # what VLBLitModule builds for LoRA: interleaved gate/up rows, SwiGLU + saved [gate | up] in the GEMM epilogue
        lw["wgu_il"] = ops.interleave_gate_up(lw["wgu"][:g.ff], lw["wgu"][g.ff:])
        del lw["wgu"]
    w.layers = [lw]
    w.final_norm = p["model.norm.weight"].to(dev, BF)
    inv = 1.0 / (g.rope_theta ** (torch.arange(0, g.head_dim, 2, dtype=torch.float32) / g.head_dim))
    fr = torch.arange(S, dtype=torch.float32)[:, None] * inv[None]
    w.rope_cos, w.rope_sin = fr.cos().to(dev).contiguous(), fr.sin().to(dev).contiguous()
    bb = Backbone(g, w)
    lora = LoraState(g, w, 16, 32, 0.1, dev, sd=p)
    hidden, _ = lora.decoder_forward(bb, x.to(dev).view(S, g.dim).clone(), mask.to(torch.uint8).to(dev), 1)
    lora.backward(bb, gout.to(dev).view(S, g.dim).contiguous())
    torch.cuda.synchronize()
    # ---- oracle with the same dropout masks (step counter is 1 after the first forward)
    drop, idx = {}, 0
    for gname, targets in GROUPS:
        for t in targets:
            K = lora.in_dims[t]
            drop[f"model.layers.0.{t}"] = (keep_mask(lora._seed(0, idx), S, K, 0.1).float() / 0.9).view(1, S, K)
            idx += 1
    names = [k for k in p if ".lora_" in k]
    pr = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in p.items()}
    ref = O.mistral_decoder(pr, x.float(), mask, go, lora_drop=drop)
    (ref * gout.float()).sum().backward()
    got = hidden.view(1, S, g.dim).float().cpu()
    err = ((got - ref.detach()).abs() * mask[..., None]).max() / ref.detach().abs().max()
    assert err < 2e-2, err
    worst = 0.0
    for n in names:
        gpu = lora.grads[n].t() if "lora_B" in n else lora.grads[n]
        e = rel_err(gpu, pr[n].grad)
        worst = max(worst, e)
        assert e < 4e-2, (n, e)
    assert len(names) == 14 and worst > 0


def test_7b_lora_step_init_properties(dev):
    """Whole 7B LoRA step (32 layers, B=1) at initialisation, where peft's B = 0 gives size-independent facts:
    every lora_A gradient is EXACTLY zero (dA = (s dY.B)^T x and B = 0), every lora_B gradient is finite and non-zero,
    the validation loss (adapters on, dropout off) equals the training loss, and both equal the frozen-backbone loss of
    the same weights to bf16 rounding (the adapters add exactly 0 to every GEMM)."""
    import warnings
    from phantom_vlb_amd.litmodule import VLBLitModule, VLBLitModuleConfig
    from phantom_vlb_amd.synthetic import synthetic_batch
    kw = dict(model_path="none", dropout_rate=0.0, num_target=2048, l2_lambda=1e-3, lr=1e-4, betas=[0.9, 0.999], eps=1e-8,
              weight_decay=1e-2, lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000, geometry="7b")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = VLBLitModule(VLBLitModuleConfig(freeze_backbone=False, use_lora=True, lora_r=16, lora_alpha=32, lora_dropout=0.1, **kw))
        m.configure_model()
    m.configure_optimizers()
    batch = synthetic_batch(m.geometry, 1, seed=5, device=m.device)
    batch["language"], batch["padvals"] = batch["language"].cpu(), batch["padvals"].cpu()
    loss = float(m.training_step(batch))
    n_a = n_b = 0
    for n, p in m.trainable_named_parameters():
        if "lora_A" in n:
            assert p.grad.count_nonzero().item() == 0, n
            n_a += 1
        elif "lora_B" in n:
            assert torch.isfinite(p.grad).all() and p.grad.abs().max() > 0, n
            n_b += 1
    assert n_a == n_b == 32 * 7
    # validation (eval mode: adapters on, dropout off) at B = 0 runs the same GEMMs with an exactly-zero adapter term
    val = float(m.validation_step(batch)["loss"])
    assert abs(val - loss) <= 1e-5 * abs(loss), (val, loss)
    # ... and the bare frozen backbone (fused SwiGLU epilogue instead of the saved gate/up) agrees to bf16 rounding
    lora, m.lora = m.lora, None
    try:
        frozen = float(m.validation_step(batch)["loss"])
    finally:
        m.lora = lora
    assert abs(frozen - loss) <= 3e-3 * abs(loss), (frozen, loss)
    del m
    torch.cuda.empty_cache()


def test_full_finetune_backward_full_size_vs_oracle(dev):
    """configs[4]'s backward at BASELINE widths: ONE 12-frame clip through 2 ViT-L layers (frozen), the FULL STC connector
    (2 x 4 RegStage blocks at 4096 channels, Conv3d 12x24x24 -> 7x13x13, readout MLP), the token splice and ONE
    Mistral-7B-sized decoder layer + final norm - every weight gradient of connector, embed_tokens (32000 x 4096) and
    decoder (wgrad GEMMs on transposed activations at M=2048, norm gains) against oracle autograd."""
    import dataclasses
    import vlb_oracle as O
    from phantom_vlb_amd.backbone import Backbone, Weights
    from phantom_vlb_amd.fullft import FullFineTune
    from phantom_vlb_amd.geometry import geometry_7b
    go = dataclasses.replace(O.geometry_7b(), layers=1, vit_layers=3)
    p = O.round_bf16(O.init_params(go, seed=21))
    g = dataclasses.replace(geometry_7b(), layers=1, vit_layers=3)
    batch = O.synthetic_batch(go, 1, seed=22)
    S = g.max_len
    gen = torch.Generator().manual_seed(23)
    gout = (torch.randn(1, S, g.dim, generator=gen) * 0.05).to(BF)
    trained = [n for n in p if n.startswith("model.") and not n.startswith("model.vision_tower.")]      # the head is not on this path
    pr = {k: (v.clone().requires_grad_(True) if k in trained else v) for k, v in p.items()}
    hid_ref, km = O.backbone_forward(pr, batch, go)
    gout = gout * km[..., None].to(BF)                     # the head never weights padded rows
    (hid_ref * gout.float()).sum().backward()
    bb = Backbone(g, Weights(g, p, dev, keep_transposed=True))
    full = FullFineTune(g, bb, dev)
    ids = batch["language"].long()
    hidden, _ = full.forward(batch["vision"].to(dev), ids.to(dev), layout=None, ids_host=ids)
    assert rel_err(hidden.view(1, S, g.dim).float().cpu() * km[..., None], hid_ref.detach() * km[..., None]) < 3e-2
    full.backward(gout.to(dev).view(S, g.dim).contiguous())
    torch.cuda.synchronize()
    grads = full.state_dict("grad")
    assert set(grads) == set(trained)
    worst = {}
    for n, gk in grads.items():
        ref = pr[n].grad
        assert ref is not None and gk.shape == ref.shape, n
        worst[n] = float((gk - ref).abs().max() / (ref.abs().max() + 1e-20))
    bad = {n: e for n, e in worst.items() if e > 8e-2}
    assert not bad, sorted(bad.items(), key=lambda t: -t[1])[:8]
    # the embedding gradient is exactly zero for tokens that do not occur in the clip
    used = set(ids[0].tolist()) - {-201}
    ge = grads["model.embed_tokens.weight"]
    unused = torch.tensor(sorted(set(range(g.vocab)) - used)[:2000])
    assert float(ge[unused].abs().max()) == 0.0
    del full, bb
    torch.cuda.empty_cache()
