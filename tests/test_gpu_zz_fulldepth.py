"""Parity at the HEADLINE configuration's full depth (runs last: the oracle side takes minutes on the host cores).

north_star: "loss matching CPU reference to 1e-3 rel".  The mini model proves that on 2 decoder layers; here the whole
VideoLLaMA2-7B geometry runs - 23 executed ViT-L/14-336 layers, the full STC connector, the splice and all 32 Mistral decoder
layers at d=4096 / ff=14336 / S=2048 - on ONE synthetic clip, and the loss and the predicted BOLD are compared with the fp32
oracle (oracle/vlb_oracle.py: training_loss, reference src/litmodule/videollama2_vlb_litmodule.py:229-306) fed the SAME
bf16-valued weights: frozen backbone (configs[1]) and LoRA r=16 with B != 0, dropout off (configs[2]) - for the latter also the
GRADIENTS the optimiser consumes (head + the 14 adapter matrices of layers 0 / 15 / 31) against the oracle's loss.backward().  This is where bf16
rounding accumulates through 23 + 32 residual layers.  Plus one whole-model configs[4] step (full-parameter fine-tune, 65,536-voxel
head, bf16 and MX-fp8 GEMMs) so the driver's GPU run exercises it.

Progress lines go to gpurun_out/fulldepth_progress.log (a run that prints nothing for minutes looks hung to the GPU pool)."""
import os
import time

import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT, rel_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
_T0 = time.time()


def _progress(msg):
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "fulldepth_progress.log"), "a") as f:
            f.write(f"[{time.time() - _T0:7.1f}s] {msg}\n")
    except OSError:
        pass


def _state_dict_7b(g, dev, seed, lora):
    """bf16 weights of the whole architecture generated on the device (7.3 B values: seconds there, minutes with a CPU
    generator), with non-trivial norm gains / biases, the brain head, and LoRA adapters with B != 0 (peft layout)."""
    from phantom_vlb_amd.backbone import Weights
    from phantom_vlb_amd.head import HEAD_PARAMS
    from phantom_vlb_amd.lora import GROUPS
    sd = Weights.random_state_dict(g, dev, seed=seed)
    gen = torch.Generator(device=dev).manual_seed(seed + 1)
    for k, t in sd.items():
        if t.dim() == 1:                                   # norm gains, biases, class embedding: make the affine parts visible
            t.add_((torch.randn(t.shape, generator=gen, device=dev) * 0.02).to(BF))
    bound = 1.0 / g.dim ** 0.5
    sd["layer_norm1.weight"] = (1 + 0.02 * torch.randn(g.dim, generator=gen, device=dev)).to(BF)
    sd["layer_norm1.bias"] = (0.02 * torch.randn(g.dim, generator=gen, device=dev)).to(BF)
    sd["layer_norm2.weight"] = (1 + 0.02 * torch.randn(g.dim, generator=gen, device=dev)).to(BF)
    sd["layer_norm2.bias"] = (0.02 * torch.randn(g.dim, generator=gen, device=dev)).to(BF)
    sd["ridge_layer.linear.weight"] = ((torch.rand(g.num_target, g.dim, generator=gen, device=dev) * 2 - 1) * bound).to(BF)
    sd["ridge_layer.linear.bias"] = ((torch.rand(g.num_target, generator=gen, device=dev) * 2 - 1) * bound).to(BF)
    assert all(n in sd for n in HEAD_PARAMS)
    if lora:
        qd, kd = g.heads * g.head_dim, g.kv_heads * g.head_dim
        dims = {"self_attn.q_proj": (qd, g.dim), "self_attn.k_proj": (kd, g.dim), "self_attn.v_proj": (kd, g.dim),
                "self_attn.o_proj": (g.dim, qd), "mlp.gate_proj": (g.ff, g.dim), "mlp.up_proj": (g.ff, g.dim), "mlp.down_proj": (g.dim, g.ff)}
        for i in range(g.layers):
            for _, targets in GROUPS:
                for t in targets:
                    out_f, in_f = dims[t]
                    sd[f"model.layers.{i}.{t}.lora_A.weight"] = ((torch.rand(g.lora_r, in_f, generator=gen, device=dev) * 2 - 1) / in_f ** 0.5).to(BF)
                    sd[f"model.layers.{i}.{t}.lora_B.weight"] = (torch.randn(out_f, g.lora_r, generator=gen, device=dev) * 0.02).to(BF)
    return sd


def test_7b_full_depth_loss_and_prediction_vs_oracle(dev):
    import vlb_oracle as O
    from phantom_vlb_amd.head import HEAD_PARAMS
    from phantom_vlb_amd.litmodule import VLBLitModule, VLBLitModuleConfig
    from phantom_vlb_amd.lora import GROUPS
    GRAD_LAYERS = (0, 15, 31)
    g = O.geometry_7b(num_target=2048, lora_r=16, lora_alpha=32)
    assert (g.vit_layers_run, g.layers, g.vis_tokens, g.max_len) == (23, 32, 1183, 2048)
    _progress("full-depth parity: generating 7.3 B bf16 weights on the device")
    sd = _state_dict_7b(g, dev, seed=31, lora=True)
    p = {k: v.to("cpu", torch.float32) for k, v in sd.items()}                 # the oracle's copy: same values, fp32 math
    p_frozen = {k: v for k, v in p.items() if ".lora_" not in k}
    batch = O.synthetic_batch(g, 1, seed=32)
    # ---- device: ONE module with the adapters; the frozen configuration is the same module with the adapters detached
    cfg = VLBLitModuleConfig(model_path="none", freeze_backbone=False, use_lora=True, lora_r=16, lora_alpha=32, lora_dropout=0.0,
                             dropout_rate=0.0, num_target=2048, l2_lambda=1e-3, lr=1e-4, betas=[0.9, 0.999], eps=1e-8,
                             weight_decay=1e-2, lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000, geometry="7b")
    m = VLBLitModule(cfg)
    m.configure_model(state_dict=sd)
    del sd
    torch.cuda.empty_cache()
    m.configure_optimizers()
    assert len(m.backbone.w.vit) == 23 and len(m.backbone.w.layers) == 32 and len(m.lora.layers) == 32
    dbatch = {k: (v.to(dev) if torch.is_tensor(v) and k not in ("language", "padvals") else v) for k, v in batch.items()}
    vid_dev = m.backbone.video_tokens(dbatch["vision"]).float().cpu()
    loss_lora = float(m.training_step(dbatch))             # dropout 0: the training forward IS the adapted forward; backward runs too
    pred_lora = m.head.pred.float().cpu().clone()
    gsq = sum(float(t.grad.float().pow(2).sum()) for _, t in m.trainable_named_parameters())
    assert gsq > 0 and gsq == gsq                          # finite, non-zero gradients through all 32 layers
    # what the optimiser consumes: head gradients and the 14 adapter gradients of the first, a middle and the last layer
    # (reference :113-120, 259-306: peft's A / B are the only backbone tensors with requires_grad)
    grad_names = list(HEAD_PARAMS) + [f"model.layers.{i}.{t}.lora_{ab}.weight" for i in GRAD_LAYERS for _, ts in GROUPS for t in ts
                                      for ab in "AB"]
    dev_grads = {}
    for n in grad_names:
        gt = m.head.grads[n] if n in m.head.grads else m.lora.grads[n]
        dev_grads[n] = (gt.t() if "lora_B" in n else gt).float().cpu().clone()
    val_lora = float(m.validation_step(dbatch)["loss"])
    lora, m.lora = m.lora, None
    try:
        out = m.validation_step(dbatch)
        loss_frozen, pred_frozen = float(out["loss"]), out["brain_preds"].float().cpu().clone()
    finally:
        m.lora = lora
    torch.cuda.synchronize()
    _progress(f"device side done: frozen loss {loss_frozen:.6f}, LoRA loss {loss_lora:.6f} (eval {val_lora:.6f})")
    del m, lora
    torch.cuda.empty_cache()
    # ---- oracle: vision side once, then the decoder + head for both weight sets
    with torch.no_grad():
        pix = batch["vision"].to(BF).float().reshape(g.num_frames, 3, g.image_size, g.image_size)        # reference :267
        vit = O.clip_tower(p, pix, g).view(1, g.num_frames, g.grid * g.grid, g.vit_dim)
        _progress("oracle: CLIP tower (23 layers) done")
        vid = O.stc_connector(p, vit, g)
        _progress("oracle: STC connector done")
        emb, km = O.splice_multimodal(p["model.embed_tokens.weight"], batch["language"].long(), vid)
        wm = O.make_weight_mask(batch["padvals"], batch["vis_weights"], batch["lang_weights"], g.lang_len, g.max_len,
                                g.ds_grid ** 2).to(BF).float()
        y = batch["timeseries"].to(BF).float()
        ref = {}
        hid = O.mistral_decoder(p_frozen, emb, km, g)
        pred, l2, _ = O.brain_head(p_frozen, hid, wm, g)
        ref["frozen"] = (float(F.mse_loss(pred, y) + l2), pred)
        _progress(f"oracle: 32 decoder layers + head (frozen) done, loss {ref['frozen'][0]:.6f}")
        del hid
    # the LoRA pass WITH autograd (host memory: ~1.5 GB of saved fp32 activations per layer; the GPU box has > 200 GB): the same
    # loss.backward() the reference's Trainer runs, for the tensors listed in grad_names
    for n in grad_names:
        p[n].requires_grad_(True)
    hid = O.mistral_decoder(p, emb, km, g)
    pred, l2, _ = O.brain_head(p, hid, wm, g)
    loss = F.mse_loss(pred, y) + l2
    ref["lora"] = (float(loss), pred.detach())
    _progress(f"oracle: 32 decoder layers + head (lora, autograd on) done, loss {ref['lora'][0]:.6f}")
    loss.backward()
    _progress("oracle: backward through 32 decoder layers done")
    del hid, loss
    worst, table = {}, []
    for n in grad_names:
        a, b = dev_grads[n], p[n].grad
        err = float((a - b).abs().max() / (b.abs().max() + 1e-30))
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        key = "head" if ".lora_" not in n else ("q/k adapters" if (".q_proj." in n or ".k_proj." in n) else "v/o/mlp adapters")
        w = worst.setdefault(key, [0.0, 1.0])
        w[0], w[1] = max(w[0], err), min(w[1], cos)
        table.append((n, err, cos))
    _progress("gradients vs oracle autograd (worst max-error / max, worst cosine): " +
              "; ".join(f"{k} {v[0]:.2e} / {v[1]:.5f}" for k, v in worst.items()))
    for n, err, cos in table:
        _progress(f"   {n}: max-error / max {err:.3e}, cosine {cos:.5f}")
    # bf16 activations, bf16 P / dS in the attention backward and bf16 backward signals against fp32 autograd.  Measured (round 4,
    # DESIGN.md section 3): head 1.4e-2 / 0.9999; v, o and MLP adapters <= 6.3e-2 / >= 0.9969 at every depth; the q and k adapters
    # carry the largest error AT EVERY DEPTH (layer 31 as much as layer 0: 8.8e-2 ... 1.3e-1 / 0.9908 ... 0.9962) - it comes from
    # dS = P o (dP - delta) in bf16 operands (a cancellation flash-attention backward kernels share), not from depth.  Bars = the
    # measured worst case with a 1.3x margin on the error and a third of the cosine deficit.
    bars = {"head": (2.5e-2, 0.9995), "v/o/mlp adapters": (8e-2, 0.995), "q/k adapters": (1.7e-1, 0.988)}
    for k, (emax, cmin) in bars.items():
        assert worst[k][0] <= emax and worst[k][1] >= cmin, (k, worst[k], table)
    assert rel_err(vid_dev.view(1, g.vis_tokens, g.dim), vid) < 6e-2           # 23 ViT layers + 8 RegStage blocks in bf16
    e_f = abs(loss_frozen - ref["frozen"][0]) / ref["frozen"][0]
    e_l = abs(loss_lora - ref["lora"][0]) / ref["lora"][0]
    p_f, p_l = rel_err(pred_frozen, ref["frozen"][1]), rel_err(pred_lora, ref["lora"][1])
    _progress(f"rel. loss error frozen {e_f:.2e} LoRA {e_l:.2e}; prediction error (max / max) frozen {p_f:.2e} LoRA {p_l:.2e}")
    assert e_f < 1e-3, (loss_frozen, ref["frozen"][0])                          # north_star: 1e-3 relative on the loss
    assert e_l < 1e-3, (loss_lora, ref["lora"][0])
    assert p_f < 3e-2 and p_l < 3e-2, (p_f, p_l)                                # predicted BOLD through 55 bf16 layers
    assert abs(val_lora - loss_lora) <= 1e-5 * abs(loss_lora)                   # eval == train forward at dropout 0
    assert abs(ref["lora"][0] - ref["frozen"][0]) > 1e-3 * ref["frozen"][0]     # the adapters (B != 0) do move the loss


def test_configs4_whole_model_step_bf16_and_fp8(dev):
    """configs[4]'s model side as ONE whole step on the GPU: full-parameter fine-tune of the 7B geometry (everything but the
    vision tower trains), 65,536-voxel head, B = 1 - bf16 GEMMs, then the same weights and clip with the decoder GEMMs on the
    MX-fp8 MFMA path: finite non-zero gradients for every trained tensor group, fp8 loss within 1e-3 of bf16, gradient cosines by depth,
    and an optimiser step that moves the weights."""
    from phantom_vlb_amd.litmodule import VLBLitModule, VLBLitModuleConfig
    from phantom_vlb_amd.synthetic import synthetic_batch
    import warnings
    _progress("configs[4] step: building the 7B full fine-tune module (65,536-voxel head)")
    cfg = VLBLitModuleConfig(model_path="none", freeze_backbone=False, use_lora=False, lora_r=None, lora_alpha=None, lora_dropout=None,
                             dropout_rate=0.0, num_target=65536, l2_lambda=1e-3, lr=1e-4, betas=[0.9, 0.999], eps=1e-8,
                             weight_decay=1e-2, lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000, geometry="7b")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = VLBLitModule(cfg)
        m.configure_model()
    opt, _ = m.configure_optimizers()
    f = m.full.flat
    batch = synthetic_batch(m.geometry, 1, seed=41, device=dev)
    batch["language"], batch["padvals"] = batch["language"].cpu(), batch["padvals"].cpu()
    loss16 = float(m.training_step(batch))
    g16 = f.grad.clone()
    head16 = m.flat.grad.clone()
    assert loss16 == loss16 and torch.isfinite(g16.float()).all() and torch.isfinite(head16).all()
    for name in ("embed_tokens", "layers.0.wqkv", "layers.15.wgu", "layers.31.wdown", "layers.31.post_norm", "norm",
                 "mm_projector.s1.b1.conv1", "mm_projector.sampler.weight", "mm_projector.ro2.weight"):
        assert float(f.g_(name).float().abs().max()) > 0, name
    assert float(m.head.grads["ridge_layer.linear.weight"].abs().max()) > 0
    _progress(f"configs[4] bf16 step done, loss {loss16:.6f}")
    m.full.fp8 = True                                      # same weights, decoder GEMMs on MX-fp8 (W / W^T quantised here)
    m.full.refresh_transposed()
    loss8 = float(m.training_step(batch))
    g8 = f.grad
    assert torch.isfinite(g8.float()).all()
    assert abs(loss8 - loss16) <= 1e-3 * abs(loss16), (loss8, loss16)      # measured 7e-5
    # gradient agreement bf16 vs MX-fp8 (e4m3 operands: ~4e-2 of the output rms per K = 4096 GEMM, tests/test_gpu_fp8.py), by
    # depth: the error of the backward signal accumulates through the fp8 dgrad GEMMs of the layers above
    cos = {}
    for name in ("layers.31.wdown", "layers.24.wqkv", "layers.15.wgu", "layers.0.wo"):
        o, k, _ = f.offsets[name]
        a, b = g16[o:o + k].float(), g8[o:o + k].float()
        cos[name] = float((a * b).sum() / (a.norm() * b.norm()))
    _progress(f"configs[4] fp8 step done, loss {loss8:.6f}, gradient cosines vs bf16 {cos}")
    assert cos["layers.31.wdown"] > 0.97 and cos["layers.15.wgu"] > 0.93 and cos["layers.0.wo"] > 0.80, cos
    w_before = f.view(f.compute, "layers.7.wo").clone()
    opt[0].step()
    torch.cuda.synchronize()
    assert not torch.equal(w_before, f.view(f.compute, "layers.7.wo"))
    assert m.optimizer.grad_norm() > 0
    del m, opt
    torch.cuda.empty_cache()
