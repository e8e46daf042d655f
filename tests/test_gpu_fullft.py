"""Full-parameter fine-tuning (BASELINE configs[4]: freeze_backbone=False, use_lora=False - everything but the vision
tower trains, reference litmodule :86-99): the HBM-bound backward pieces against torch autograd, and one whole mini
training step - every gradient of connector, embeddings, decoder and head - against the oracle's autograd."""
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _r(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def test_transpose_pad(dev):
    from phantom_vlb_amd import ops
    for (R, C, ld) in [(100, 64, 64), (5861, 1024, 3072), (77, 200, 200), (64, 8, 8)]:
        full = _r(R, ld, seed=R).to(dev)
        x = full[:, :C]
        Rp = (R + 63) // 64 * 64
        out = torch.full((C, Rp + 64), 7.0, dtype=BF, device=dev)
        ops.transpose_pad(x, out, Rp)
        assert torch.equal(out[:, :R], x.t())
        assert float(out[:, R:Rp].abs().max()) == 0.0 if Rp > R else True
        assert float((out[:, Rp:] - 7).abs().max()) == 0.0          # beyond the padded width: untouched


def test_wgrad_is_the_tn_gemm_on_transposed_activations(dev):
    from phantom_vlb_amd import ops
    for (M, N, K) in [(300, 256, 128), (5861, 512, 256), (36, 256, 64)]:
        dy, x = _r(M, N, seed=1).to(dev), _r(M, K, seed=2).to(dev)
        Mp = (M + 63) // 64 * 64
        dyT, xT = torch.empty(N, Mp, dtype=BF, device=dev), torch.empty(K, Mp, dtype=BF, device=dev)
        ops.transpose_pad(dy, dyT, Mp); ops.transpose_pad(x, xT, Mp)
        dW = ops.gemm(dyT, xT)
        ref = dy.float().t() @ x.float()
        assert rel_err(dW, ref) < 1e-2, (M, N, K)


@pytest.mark.parametrize("rows,dim", [(70, 64), (5861, 4096), (1000, 1024)])
def test_rmsnorm_bwd_dw(dev, rows, dim):
    from phantom_vlb_amd import ops
    x, dy = _r(rows, dim, seed=3), _r(rows, dim, seed=4)
    xf = x.float()
    ref = (dy.float() * xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)).sum(0)
    out = torch.empty(dim, dtype=BF, device=dev)
    ops.rmsnorm_bwd_dw(x.to(dev), dy.to(dev), 1e-5, out)
    assert rel_err(out, ref) < 1e-2
    # the fused form: same d gamma, and dx as the separate kernel's (with and without the residual gradient)
    w, dxi = (1 + 0.1 * _r(dim, seed=5).float()).to(BF).to(dev), _r(rows, dim, seed=6).to(dev)
    for res in (None, dxi):
        out2 = torch.empty(dim, dtype=BF, device=dev)
        dx = ops.rmsnorm_bwd_full(x.to(dev), w, dy.to(dev), 1e-5, out2, dx_in=res)
        assert rel_err(out2, ref) < 1e-2          # partial sums are grouped differently from the separate kernel's
        assert rel_err(dx, ops.rmsnorm_bwd(x.to(dev), w, dy.to(dev), 1e-5, dx_in=res)) < 4e-3


@pytest.mark.parametrize("rows,dim,act,res", [(50, 64, 3, True), (20736, 4096, 3, True), (777, 1024, 0, False), (300, 512, 3, False),
                                               (128, 4096, 0, True)])
def test_layernorm_bwd_matches_autograd(dev, rows, dim, act, res):
    """y = act(LN(x) + residual): dx, d residual, dw, db against autograd on the same bf16-valued inputs."""
    from phantom_vlb_amd import ops
    x, w, b = _r(rows, dim, seed=5), (1 + 0.1 * _r(dim, seed=6).float()).to(BF), _r(dim, seed=7, scale=0.1)
    r = _r(rows, dim, seed=8) if res else None
    dy = _r(rows, dim, seed=9)
    xs, ws, bs = (t.float().requires_grad_(True) for t in (x, w, b))
    rs = r.float().requires_grad_(True) if res else None
    z = F.layer_norm(xs, (dim,), ws, bs, 1e-6)
    if res:
        z = z + rs
    y = F.silu(z) if act == 3 else z
    y.backward(dy.float())
    dw, db = torch.empty(dim, dtype=BF, device=dev), torch.empty(dim, dtype=BF, device=dev)
    dx, dres = ops.layernorm_bwd(x.to(dev), w.to(dev), b.to(dev), dy.to(dev), 1e-6, dw, db, residual=None if r is None else r.to(dev),
                                 act=act, want_dres=res)
    assert rel_err(dx, xs.grad) < 1.5e-2
    assert rel_err(dw, ws.grad) < 1.5e-2 and rel_err(db, bs.grad) < 1.5e-2
    if res:
        assert rel_err(dres, rs.grad) < 1.5e-2
    # the forward this is the backward of
    yk = ops.layernorm(x.to(dev), w.to(dev), b.to(dev), 1e-6, residual=None if r is None else r.to(dev), act=act)
    assert rel_err(yk, y.detach()) < 1e-2


def test_act_fwd_bwd_colsum(dev):
    from phantom_vlb_amd import ops
    x, dy = _r(333, 256, seed=1), _r(333, 256, seed=2)
    for act, fn in ((ops.ACT_SILU, F.silu), (ops.ACT_GELU, F.gelu)):
        xs = x.float().requires_grad_(True)
        y = fn(xs)
        y.backward(dy.float())
        assert rel_err(ops.act_fwd(x.to(dev), act), y.detach()) < 1e-2
        assert rel_err(ops.act_bwd(x.to(dev), dy.to(dev), act), xs.grad) < 1e-2
    out = torch.empty(256, dtype=BF, device=dev)
    ops.colsum(dy.to(dev), out)
    assert rel_err(out, dy.float().sum(0)) < 1e-2
    wide = _r(5861, 512, seed=3).to(dev)
    out2 = torch.empty(128, dtype=BF, device=dev)
    ops.colsum(wide[:, 64:192], out2)                        # a column slice: row stride respected
    assert rel_err(out2, wide[:, 64:192].float().sum(0)) < 1e-2


def test_embed_grad_sums_rows_per_token(dev):
    from phantom_vlb_amd import ops
    d = _r(50, 64, seed=4).to(dev)
    tok = torch.tensor([3, 7, 9], dtype=torch.int32, device=dev)
    beg = torch.tensor([0, 2, 3, 6], dtype=torch.int32, device=dev)
    rows = torch.tensor([1, 40, 5, 0, 10, 49], dtype=torch.int32, device=dev)
    out = torch.zeros(12, 64, dtype=BF, device=dev)
    ops.embed_grad(d, tok, beg, rows, out, 64)
    df = d.float()
    assert rel_err(out[3], df[1] + df[40]) < 1e-2 and rel_err(out[7], df[5]) < 1e-2
    assert rel_err(out[9], df[0] + df[10] + df[49]) < 1e-2
    assert float(out[[0, 1, 2, 4, 5, 6, 8, 10, 11]].abs().max()) == 0.0


def test_dwconv_se_col2im_backward_pieces(dev):
    from phantom_vlb_amd import ops
    N, H, C = 3, 6, 64
    x, dy = _r(N * H * H, C, seed=1), _r(N * H * H, C, seed=2)
    w = _r(C, 1, 3, 3, seed=3, scale=0.3)
    xs = x.float().view(N, H, H, C).permute(0, 3, 1, 2).requires_grad_(True)
    ws = w.float().requires_grad_(True)
    y = F.conv2d(xs, ws, padding=1, groups=C)
    y.backward(dy.float().view(N, H, H, C).permute(0, 3, 1, 2))
    w9 = w.flatten(1).t().contiguous().to(dev)                        # [9, C] tap-major, the kernels' layout
    dw = torch.empty(9, C, dtype=BF, device=dev)
    ops.dwconv3x3_bwd_w(x.to(dev), dy.to(dev), N, H, H, C, dw)
    assert rel_err(dw, ws.grad.flatten(1).t()) < 1e-2
    dx = ops.dwconv3x3(dy.to(dev), torch.flip(w9, dims=[0]).contiguous(), N, H, H, C)
    assert rel_err(dx, xs.grad.permute(0, 2, 3, 1).reshape(N * H * H, C)) < 1e-2
    # squeeze-excite: y = x * sigmoid(s), s a function of mean(x) upstream
    s, dp = _r(N, C, seed=4), _r(N, C, seed=5)
    xs2 = x.float().view(N, H * H, C).requires_grad_(True)
    ss = s.float().requires_grad_(True)
    out = xs2 * torch.sigmoid(ss)[:, None, :]
    pooled = xs2.mean(1)
    (out * dy.float().view(N, H * H, C)).sum().backward(retain_graph=True)
    ds = ops.se_bwd_gate(x.to(dev), dy.to(dev), s.to(dev), N, H * H, C)
    assert rel_err(ds, ss.grad) < 1e-2
    gx = xs2.grad.clone()
    xs2.grad = None
    (pooled * dp.float()).sum().backward()
    dxk = ops.se_bwd_x(dy.to(dev), s.to(dev), dp.to(dev), N, H * H, C)
    assert rel_err(dxk, (gx + xs2.grad).reshape(N * H * H, C)) < 1e-2
    # col2im3d is the exact inverse of im2col3d on real (non-padding) elements
    B, T, G = 2, 8, 6
    v = _r(B * T * G * G, C, seed=6).to(dev)
    cols = ops.im2col3d(v, B, T, G, G, C)
    assert torch.equal(ops.col2im3d(cols.view(-1, 8 * C), B, T, G, G, C), v)


def test_adamw_on_bf16_gradients_matches_the_fp32_kernel(dev):
    import ctypes
    from phantom_vlb_amd._lib import check, lib
    n = 8 * 1000
    g = _r(n, seed=1).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for g16 in (False, True):
        p = torch.linspace(-1, 1, n, device=dev)
        pb = torch.empty(n, dtype=BF, device=dev)
        m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        ss, ws = torch.zeros(1, device=dev), torch.zeros(2048, device=dev)
        gg = g if g16 else g.float()
        for step in (1, 2, 3):
            ss.zero_()
            check((lib.vlb_grad_sumsq_bf16 if g16 else lib.vlb_grad_sumsq)(gg.data_ptr(), n, ss.data_ptr(), ws.data_ptr(), st), "sumsq")
            check((lib.vlb_adamw_step_g16 if g16 else lib.vlb_adamw_step)(p.data_ptr(), pb.data_ptr(), gg.data_ptr(), m.data_ptr(), v.data_ptr(),
                                                                        n, 1e-3, 0.9, 0.999, 1e-8, 1e-2, step, ss.data_ptr(), 1.0, st), "adamw")
        torch.cuda.synchronize()
        outs.append((p.clone(), pb.clone(), float(ss)))
    assert abs(outs[0][2] - outs[1][2]) / outs[0][2] < 1e-5
    assert torch.allclose(outs[0][0], outs[1][0], atol=1e-6) and torch.equal(outs[0][1], outs[1][1])


def _cfg(**kw):
    from phantom_vlb_amd.litmodule import VLBLitModuleConfig
    base = dict(model_path="none", freeze_backbone=False, use_lora=False, lora_r=None, lora_alpha=None, lora_dropout=None,
                dropout_rate=0.0, num_target=128, l2_lambda=1e-3, lr=1e-3, betas=[0.9, 0.999], eps=1e-8, weight_decay=1e-2,
                lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000, geometry="mini")
    base.update(kw)
    return VLBLitModuleConfig(**base)


@pytest.mark.parametrize("pack", [False, True])
def test_mini_full_finetune_gradients_match_oracle_autograd(dev, pack):
    """configs[4] in miniature: every trained tensor outside the vision tower - connector (RegStage x2, Conv3d, readout),
    embed_tokens, decoder linears and norms, final norm, head - against torch autograd through the oracle."""
    import vlb_oracle as O
    from phantom_vlb_amd.litmodule import VLBLitModule
    g = O.geometry_mini()
    p = O.round_bf16(O.init_params(g, seed=11))
    batch = O.synthetic_batch(g, 4, seed=12)
    trained = [n for n in p if not n.startswith("model.vision_tower.")]
    q = {k: (v.clone().requires_grad_(True) if k in trained else v) for k, v in p.items()}
    loss_ref, _ = O.training_loss(q, batch, g)
    loss_ref.backward()
    m = VLBLitModule(_cfg(pack_tokens=pack))
    m.configure_model(state_dict=p)
    opt, _ = m.configure_optimizers()
    loss = m.training_step(batch)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_ref)) / float(loss_ref) < 1e-3
    grads = m.full.state_dict("grad")
    assert set(grads) == {n for n in trained if not n.startswith(("layer_norm", "ridge_layer"))}
    worst = {}
    for n, gk in grads.items():
        ref = q[n].grad
        assert ref is not None and gk.shape == ref.shape, n
        worst[n] = float((gk - ref).abs().max() / (ref.abs().max() + 1e-12))
    bad = {n: e for n, e in worst.items() if e > 6e-2}
    assert not bad, sorted(bad.items(), key=lambda t: -t[1])[:8]
    for n in ("layer_norm1.weight", "ridge_layer.linear.weight"):
        assert rel_err(m.head.grads[n], q[n].grad) < 3e-2
    # one optimiser step: every trained tensor moves, the W^T copies follow, and the loss on the same batch drops
    before = m.full.flat.master.clone()
    opt[0].step()
    torch.cuda.synchronize()
    assert float((m.full.flat.master - before).abs().max()) > 0
    lw = m.backbone.w.layers[0]
    assert torch.equal(lw["wqkv_t"], lw["wqkv"].t()) and lw["wqkv"].data_ptr() == m.full.flat.view(m.full.flat.compute, "layers.0.wqkv").data_ptr()
    losses = [float(loss)]
    for _ in range(3):
        losses.append(float(m.training_step(batch)))
        opt[0].step()
    assert losses[-1] < losses[0]
    # validation uses the inference forward on the updated weights
    out = m.validation_step(batch)
    assert torch.isfinite(out["brain_preds"]).all()
