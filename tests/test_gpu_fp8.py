"""Block-scaled fp8 (MX: e4m3 + E8M0 per 32 K elements) MFMA GEMM, BASELINE configs[4]: the quantiser bit-for-bit
against a torch restatement of the OCP MX rule, the GEMM against an fp32 matmul of the dequantised operands, and the
parity ladder fp32 -> bf16 MFMA -> fp8 MFMA on decoder-sized operands."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _quant_ref(x):
    """x fp32 [R, K] (bf16-valued) -> (uint8 e4m3 bytes, uint8 scales, dequantised fp32)."""
    R, K = x.shape
    xb = x.view(R, K // 32, 32)
    amax = xb.abs().amax(-1)
    E = torch.where(amax > 0, torch.ceil(torch.log2(amax.double() / 448.0)).clamp(-127, 127), torch.full_like(amax, -127).double()).float()
    scaled = xb * torch.exp2(-E)[..., None]
    q = scaled.to(torch.float8_e4m3fn)
    deq = (q.float() * torch.exp2(E)[..., None]).view(R, K)
    return q.view(torch.uint8).view(R, K), (E + 127).to(torch.uint8), deq


def test_quantize_mxfp8_bit_exact(dev):
    from phantom_vlb_amd import ops
    torch.manual_seed(0)
    x = (torch.randn(300, 256) * torch.logspace(-6, 3, 300)[:, None]).to(BF)
    x[5] = 0                                     # an all-zero row: scale byte 0, zero elements
    x[6, :32] = 448.0                            # exactly representable extremes
    x[7, 3] = 2.0 ** -20                         # far below the block's resolution: flushes to zero
    qr, sr, _ = _quant_ref(x.float())
    q, s = ops.quantize_mxfp8(x.to(dev))
    assert torch.equal(s.cpu(), sr)
    qc = q.cpu()
    same = (qc == qr) | ((qc & 0x7f) == 0) & ((qr & 0x7f) == 0)       # +0 / -0 are the same value
    assert bool(same.all()), int((~same).sum())


def test_transpose_quantize_equals_quantize_of_the_transpose(dev):
    from phantom_vlb_amd import ops
    torch.manual_seed(3)
    for R, C, ld in ((300, 200, 256), (5861, 1024, 1024), (64, 72, 72), (128, 64, 64)):
        full = (torch.randn(R, ld) * torch.logspace(-3, 2, R)[:, None]).to(BF).to(dev)
        x = full[:, :C]
        Rp = (R + 127) // 128 * 128
        q = torch.full((C, Rp), 0x55, dtype=torch.uint8, device=dev); s = torch.full((C, Rp // 32), 0x55, dtype=torch.uint8, device=dev)
        ops.transpose_quantize_mxfp8(x, q, s, Rp)
        xt = torch.zeros(C, Rp, dtype=BF, device=dev); xt[:, :R] = x.t()
        q2, s2 = ops.quantize_mxfp8(xt)
        assert torch.equal(s, s2) and torch.equal(q, q2), (R, C)


def test_quantize_dual_equals_the_two_separate_kernels(dev):
    """vlb_quantize_dual_mxfp8: row-wise + transposed quantisation of a backward signal from ONE read - both outputs bit-identical
    to vlb_quantize_mxfp8 / vlb_transpose_quantize_mxfp8 (ragged row counts, a column slice of a wider tensor)."""
    from phantom_vlb_amd import ops
    torch.manual_seed(5)
    for R, C, ld in ((300, 192, 256), (5861, 1024, 1024), (77, 64, 64), (128, 6144, 6144)):
        full = (torch.randn(R, ld) * torch.logspace(-3, 2, R)[:, None]).to(BF).to(dev)
        x = full[:, :C]
        Rp = (R + 127) // 128 * 128
        q = torch.full((R, C), 0x55, dtype=torch.uint8, device=dev); s = torch.full((R, C // 32), 0x55, dtype=torch.uint8, device=dev)
        qt = torch.full((C, Rp), 0x55, dtype=torch.uint8, device=dev); st = torch.full((C, Rp // 32), 0x55, dtype=torch.uint8, device=dev)
        ops.quantize_dual_mxfp8(x, q, s, qt, st, Rp)
        q1, s1 = ops.quantize_mxfp8(x)
        qt1 = torch.empty_like(qt); st1 = torch.empty_like(st)
        ops.transpose_quantize_mxfp8(x, qt1, st1, Rp)
        assert torch.equal(q, q1) and torch.equal(s, s1), (R, C)
        assert torch.equal(qt, qt1) and torch.equal(st, st1), (R, C)


def test_producers_emit_the_quantisation_of_their_output(dev):
    """vlb_rmsnorm_fwd_mxfp8 / vlb_swiglu_fwd_mxfp8: same bf16 output as the plain kernels, and q / scales bit-identical to
    quantising that output afterwards."""
    from phantom_vlb_amd import ops
    torch.manual_seed(6)
    for rows, dim in ((301, 4096), (77, 256), (5861, 1024)):
        x = (torch.randn(rows, dim) * torch.logspace(-2, 2, rows)[:, None]).to(BF).to(dev)
        w = (1 + 0.1 * torch.randn(dim)).to(BF).to(dev)
        y, q, s = ops.rmsnorm_mxfp8(x, w, 1e-5)
        y0 = ops.rmsnorm(x, w, 1e-5)
        q0, s0 = ops.quantize_mxfp8(y0)
        assert torch.equal(y, y0) and torch.equal(q, q0) and torch.equal(s, s0), (rows, dim)
    for rows, ff in ((301, 14336), (77, 64), (1000, 352)):
        gu = (torch.randn(rows, 2 * ff) * 2).to(BF).to(dev)
        h, q, s = ops.swiglu_mxfp8(gu)
        h0 = ops.swiglu(gu)
        q0, s0 = ops.quantize_mxfp8(h0)
        assert torch.equal(h, h0) and torch.equal(q, q0) and torch.equal(s, s0), (rows, ff)


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 512, 1024), (5861, 1024, 4096), (77, 256, 256),
                                   (5861, 4096, 512),      # 368 tiles: one full round + 112 tiles re-cut into 256x128 halves
                                   (3000, 3072, 256)])     # 144 tiles: everything runs as halves
def test_gemm_mxfp8_equals_fp32_matmul_of_the_dequantised_operands(dev, M, N, K):
    from phantom_vlb_amd import ops
    torch.manual_seed(1)
    a = (torch.randn(M, K) * torch.rand(M, 1).mul(4).exp()).to(BF)          # rows with very different magnitudes
    w = (torch.randn(N, K) * 0.05).to(BF)
    r = torch.randn(M, N).to(BF)
    aq, sa = ops.quantize_mxfp8(a.to(dev))
    wq, sw = ops.quantize_mxfp8(w.to(dev))
    _, _, ad = _quant_ref(a.float())
    _, _, wd = _quant_ref(w.float())
    ref = ad.double() @ wd.double().t()
    out = ops.gemm_mxfp8(aq, sa, wq, sw)
    assert rel_err(out, ref.float()) < 6e-3                                   # bf16 output rounding only
    out_r = ops.gemm_mxfp8(aq, sa, wq, sw, residual=r.to(dev))
    assert rel_err(out_r, (ref + r.double()).float()) < 6e-3


def test_parity_ladder_fp32_bf16_fp8(dev):
    """Same operands through fp32 matmul, the bf16 MFMA GEMM and the MX-fp8 MFMA GEMM: relative error (vs max |ref|)
    of the bf16 path ~1e-3 (output rounding), of the fp8 path a few 1e-2 of the output RMS - the ladder DESIGN.md quotes."""
    from phantom_vlb_amd import ops
    torch.manual_seed(2)
    M, N, K = 2048, 4096, 4096
    a = torch.randn(M, K).to(BF).to(dev)
    w = (torch.randn(N, K) * 0.02).to(BF).to(dev)
    ref = a.float() @ w.float().t()
    y16 = ops.gemm(a, w).float()
    aq, sa = ops.quantize_mxfp8(a)
    wq, sw = ops.quantize_mxfp8(w)
    y8 = ops.gemm_mxfp8(aq, sa, wq, sw).float()
    rms = float(ref.pow(2).mean().sqrt())
    e16 = float((y16 - ref).pow(2).mean().sqrt()) / rms
    e8 = float((y8 - ref).pow(2).mean().sqrt()) / rms
    print(f"ladder: bf16 rms err {e16:.2e}, mxfp8 rms err {e8:.2e}")
    assert e16 < 5e-3 and e16 < e8 < 6e-2


def test_full_finetune_step_on_the_fp8_path_tracks_the_bf16_path(dev):
    """Model-level rung of the ladder (configs[4] in miniature): the same full fine-tuning step with the decoder's
    forward / dgrad GEMMs on MX-fp8 - loss within 2 % of the bf16 path's and of the fp32 oracle's, gradients
    correlated > 0.97 with the bf16 path's, training still reduces the loss."""
    import vlb_oracle as O
    from phantom_vlb_amd.litmodule import VLBLitModule, VLBLitModuleConfig
    g = O.geometry_mini()
    p = O.round_bf16(O.init_params(g, seed=11))
    batch = O.synthetic_batch(g, 4, seed=12)
    with torch.no_grad():
        loss_ref, _ = O.training_loss(p, batch, g)
    res = {}
    for fp8 in (False, True):
        cfg = VLBLitModuleConfig(model_path="none", freeze_backbone=False, use_lora=False, lora_r=None, lora_alpha=None, lora_dropout=None,
                                 dropout_rate=0.0, num_target=128, l2_lambda=1e-3, lr=1e-3, betas=[0.9, 0.999], eps=1e-8, weight_decay=1e-2,
                                 lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000, geometry="mini", fp8_gemm=fp8)
        m = VLBLitModule(cfg)
        m.configure_model(state_dict=p)
        opt, _ = m.configure_optimizers()
        loss = float(m.training_step(batch))
        grad = m.full.flat.grad.float().clone()
        losses = [loss]
        for _ in range(3):
            opt[0].step()
            losses.append(float(m.training_step(batch)))
        res[fp8] = (loss, grad, losses)
    l16, g16, _ = res[False]
    l8, g8, tr8 = res[True]
    assert abs(l8 - l16) / l16 < 2e-2 and abs(l8 - float(loss_ref)) / float(loss_ref) < 2e-2
    cos = float((g8 * g16).sum() / (g8.norm() * g16.norm()))
    print(f"fp8 ladder: loss bf16 {l16:.5f} fp8 {l8:.5f} oracle {float(loss_ref):.5f}; grad cosine {cos:.4f}; fp8 losses {tr8}")
    assert cos > 0.97
    assert tr8[-1] < tr8[0]
