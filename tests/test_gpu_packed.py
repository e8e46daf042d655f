"""Packed (unpadded) token rows == dense rows.

The reference's decoder runs flash-attn's varlen path on unpadded tokens (modeling_mistral.py
_upad_input) - the padded tail of each clip is never computed.  Here that is the packed RowLayout;
these tests pin it to the dense layout: every kept row must come out BIT-IDENTICAL (per-row ops and
per-clip tile boundaries do not move), and the loss / head gradients likewise.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _lens_layout(ops, B, S, lens, dev):
    return ops.RowLayout(B, S, lens, device=dev)


def _pack(x_dense, lens):
    """[B,S,C] -> packed [sum(lens), C]"""
    return torch.cat([x_dense[b, :n] for b, n in enumerate(lens)], 0).contiguous()


@pytest.mark.parametrize("B,S,Hq,Hkv,lens", [(3, 160, 4, 1, [160, 37, 129]), (2, 300, 8, 2, [1, 300]),
                                             (4, 96, 2, 2, [64, 65, 96, 33])])
def test_attention_packed_equals_dense(dev, B, S, Hq, Hkv, lens):
    from phantom_vlb_amd import ops
    D = 128
    g = torch.Generator().manual_seed(sum(lens))
    qkv = (torch.randn(B, S, (Hq + 2 * Hkv) * D, generator=g) * 0.7).to(BF)
    dout = torch.randn(B, S, Hq * D, generator=g).to(BF)
    mask = torch.zeros(B, S, dtype=torch.uint8)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
        if n > 8:
            mask[b, 3] = 0                      # an interior masked key survives packing
        dout[b, n:] = 0
    qd, kd = Hq * D, Hkv * D
    dq = qkv.view(B * S, -1).to(dev)
    out_d, lse_d = ops.attention_fwd(dq[:, :qd], dq[:, qd:qd + kd], dq[:, qd + kd:], B, S, Hq, Hkv, D, True, D ** -0.5,
                                     key_mask=mask.to(dev), need_lse=True)
    dqkv_d = ops.attention_bwd(dq, qd, kd, out_d, dout.view(B * S, -1).to(dev), lse_d, mask.to(dev), B, S, Hq, Hkv, D,
                               True, D ** -0.5)
    lay = _lens_layout(ops, B, S, lens, dev)
    pq = _pack(qkv, lens).to(dev)
    pmask = _pack(mask[..., None], lens).view(-1).to(dev)
    out_p, lse_p = ops.attention_fwd(pq[:, :qd], pq[:, qd:qd + kd], pq[:, qd + kd:], B, S, Hq, Hkv, D, True, D ** -0.5,
                                     key_mask=pmask, need_lse=True, layout=lay)
    dqkv_p = ops.attention_bwd(pq, qd, kd, out_p, _pack(dout, lens).to(dev), lse_p, pmask, B, S, Hq, Hkv, D, True,
                               D ** -0.5, layout=lay)
    assert out_p.shape[0] == sum(lens)
    assert torch.equal(out_p, _pack(out_d.view(B, S, -1), lens))
    for b, n in enumerate(lens):
        assert torch.equal(lse_p[b, :, :n], lse_d[b, :, :n])
    # dq, dk, dv: every reduction runs in a fixed order relative to the clip's first row -> identical
    assert torch.equal(dqkv_p, _pack(dqkv_d.view(B, S, -1), lens))


def test_rope_and_splice_packed(dev):
    from oracle import vlb_oracle as O
    from phantom_vlb_amd import ops
    g = O.geometry_mini()
    B = 3
    batch = O.synthetic_batch(g, B, seed=5)
    ids = batch["language"].long().contiguous()
    gen = torch.Generator().manual_seed(2)
    emb = torch.randn(g.vocab, g.dim, generator=gen).to(BF)
    vid = torch.randn(B, g.vis_tokens, g.dim, generator=gen).to(BF)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    S = g.max_len
    dense, dmask = ops.splice_embed(ids.to(dev), emb.to(dev), vid.to(dev).view(-1, g.dim), g.vis_tokens, O.VIDEO_TOKEN_ID, err)
    lens = [int(S - p) for p in batch["padvals"][:, 0]]
    lay = ops.RowLayout(B, S, lens, device=dev)
    packed, pmask = ops.splice_embed(ids.to(dev), emb.to(dev), vid.to(dev).view(-1, g.dim), g.vis_tokens,
                                     O.VIDEO_TOKEN_ID, err, layout=lay)
    assert int(err.item()) == 0
    assert torch.equal(packed, _pack(dense.view(B, S, -1), lens))
    assert torch.equal(pmask, _pack(dmask[..., None], lens).view(-1))
    assert torch.equal(lay.pos.cpu(), torch.cat([torch.arange(n, dtype=torch.int32) for n in lens]))
    assert torch.equal(lay.unpack(packed)[1, :lens[1]], dense.view(B, S, -1)[1, :lens[1]])
    # rotary positions follow layout.pos
    H, D = 2, 128
    inv = 1.0 / (g.rope_theta ** (torch.arange(0, D, 2).float() / D))
    fr = torch.outer(torch.arange(S).float(), inv)
    cos, sin = fr.cos(), fr.sin()
    x = torch.randn(B, S, H * D, generator=gen).to(BF)
    xd = x.view(B * S, -1).clone().to(dev)
    ops.rope_(xd, cos.contiguous().to(dev), sin.contiguous().to(dev), B, S, H, D)
    xp = _pack(x, lens).to(dev)
    ops.rope_(xp, cos.contiguous().to(dev), sin.contiguous().to(dev), B, S, H, D, pos=lay.pos)
    assert torch.equal(xp, _pack(xd.view(B, S, -1), lens))


@pytest.mark.parametrize("E,V", [(512, 128), (4096, 2048)])
def test_head_packed_equals_dense(dev, E, V):
    from phantom_vlb_amd import ops
    from phantom_vlb_amd.head import BrainHead
    B, S = 3, 200
    lens = [200, 61, 130]
    gen = torch.Generator().manual_seed(E)
    hidden = (torch.randn(B, S, E, generator=gen) * 2).to(BF)
    wm = torch.rand(B, S, generator=gen) * 0.1
    for b, n in enumerate(lens):
        wm[b, max(n - 4, 0):] = 0              # the reference's mask: 4 + pad_len trailing zeros
        wm[b, :7] = 0
    y = torch.randn(B, V, generator=gen)
    head = BrainHead(E, V, 1e-3, 1e-5, dev, seed=3)
    pred_d, terms_d = head.forward(hidden.view(B * S, E).to(dev), wm.to(dev), y.to(dev))
    pred_d, terms_d = pred_d.clone(), terms_d.clone()
    dh_d = head.backward(need_dhidden=True)
    grads_d = {k: v.clone() for k, v in head.grads.items()}
    lay = ops.RowLayout(B, S, lens, device=dev)
    pred_p, terms_p = head.forward(_pack(hidden, lens).to(dev), wm.to(dev), y.to(dev), layout=lay)
    assert torch.equal(pred_p, pred_d) and torch.equal(terms_p, terms_d)
    dh_p = head.backward(need_dhidden=True)
    assert dh_p.shape[0] == sum(lens)
    assert torch.equal(dh_p, _pack(dh_d.view(B, S, E), lens))
    for k in grads_d:
        assert torch.equal(head.grads[k], grads_d[k]), k
    with pytest.raises(ValueError):
        head.forward(hidden.view(B * S, E).to(dev), wm.to(dev), y.to(dev), layout=lay)     # dense rows, packed layout


def _cfg(**kw):
    from phantom_vlb_amd.litmodule import VLBLitModuleConfig
    base = dict(model_path="none", freeze_backbone=True, use_lora=False, lora_r=None, lora_alpha=None,
                lora_dropout=None, dropout_rate=0.0, num_target=128, l2_lambda=1e-3, lr=1e-4, betas=[0.9, 0.999],
                eps=1e-8, weight_decay=1e-2, lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000,
                geometry="mini")
    base.update(kw)
    return VLBLitModuleConfig(**base)


def test_row_layout_from_host_ids(dev):
    from oracle import vlb_oracle as O
    from phantom_vlb_amd.litmodule import VLBLitModule
    g = O.geometry_mini()
    batch = O.synthetic_batch(g, 4, seed=9)
    m = VLBLitModule(_cfg())
    m.configure_model()
    lay = m.backbone.row_layout(batch["language"], batch["padvals"])
    S = g.max_len
    assert lay.lens == [int(S - p) for p in batch["padvals"][:, 0]]
    assert lay.rows == sum(lay.lens) and lay.smax == max(lay.lens)
    assert m.backbone.row_layout(batch["language"].to(dev)) is None        # device ids: no sync, dense layout
    # padvals claiming LESS padding than the ids show keeps the longer length (never drops a weighted row)
    pv = batch["padvals"].clone()
    pv[0, 0] = 0
    assert m.backbone.row_layout(batch["language"], pv).lens[0] == S


@pytest.mark.parametrize("lora", [False, True])
def test_training_step_packed_equals_dense(dev, lora):
    """Whole step on the mini geometry: packed and dense layouts give the same loss (bit-identical for
    the frozen path) and the same gradients (LoRA: fp32 reductions over rows regroup, so ~1e-3)."""
    from oracle import vlb_oracle as O
    from phantom_vlb_amd.backbone import Weights
    from phantom_vlb_amd.litmodule import VLBLitModule
    g = O.geometry_mini(lora_r=16, lora_alpha=32) if lora else O.geometry_mini()
    batch = O.synthetic_batch(g, 3, seed=21)
    kw = dict(freeze_backbone=False, use_lora=True, lora_r=16, lora_alpha=32, lora_dropout=0.0) if lora else {}
    out = {}
    for pack in (False, True):
        m = VLBLitModule(_cfg(pack_tokens=pack, **kw))
        m.configure_model()
        m.configure_optimizers()
        loss = m.training_step(batch)
        grads = {n: p.grad.clone() for n, p in m.trainable_named_parameters()}
        out[pack] = (float(loss), grads)
    if not lora:
        assert out[True][0] == out[False][0]
    else:
        assert abs(out[True][0] - out[False][0]) <= 1e-6 * abs(out[False][0])
    for n, gd in out[False][1].items():
        gp = out[True][1][n]
        if not lora:
            assert torch.equal(gp, gd), n
        else:
            denom = gd.abs().max().clamp_min(1e-12)
            assert (gp - gd).abs().max() / denom < 5e-3, n


def test_packed_step_edge_batches(dev):
    """Edge batches: a single clip, a clip without any padding (len == max_len), and one with the largest padding
    the generator produces - packed == dense for loss and head gradients in every case."""
    from oracle import vlb_oracle as O
    from phantom_vlb_amd.litmodule import VLBLitModule
    g = O.geometry_mini()
    S = g.max_len
    for B, seed in ((1, 3), (2, 4), (4, 5)):
        batch = O.synthetic_batch(g, B, seed=seed)
        # force clip 0 to carry no padding at all: fill its padded tail with ordinary token ids
        ids = batch["language"].clone()
        pad0 = int(batch["padvals"][0, 0])
        if pad0:
            ids[0, ids.shape[1] - pad0:] = 7.0
        batch = dict(batch, language=ids)
        res = {}
        for pack in (False, True):
            m = VLBLitModule(_cfg(pack_tokens=pack))
            m.configure_model()
            m.configure_optimizers()
            if pack:
                lay = m.backbone.row_layout(batch["language"], batch["padvals"])
                assert lay.lens[0] == S                     # nothing to drop for clip 0
            res[pack] = (float(m.training_step(batch)), m.flat.grad.clone())
        assert res[True][0] == res[False][0]
        assert torch.equal(res[True][1], res[False][1])
