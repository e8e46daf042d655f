"""configs[0] end to end on the GPU: mini VideoLLaMA2 + 128-voxel head vs committed golden vectors
and vs the live oracle (same seeded weights and clips).  -m gpu; all compute through the C-ABI."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cfg(lora=False, **kw):
    from phantom_vlb_amd.litmodule import VLBLitModuleConfig
    base = dict(model_path="none", freeze_backbone=not lora, use_lora=lora, lora_r=16 if lora else None,
                lora_alpha=32 if lora else None, lora_dropout=0.0 if lora else None, dropout_rate=0.0,
                num_target=128, l2_lambda=1e-3, lr=1e-4, betas=[0.9, 0.999], eps=1e-8, weight_decay=1e-2,
                lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000, geometry="mini")
    base.update(kw)
    return VLBLitModuleConfig(**base)


@pytest.fixture(scope="module")
def mini_frozen(dev):
    import vlb_oracle as O
    from phantom_vlb_amd.litmodule import VLBLitModule
    g = O.geometry_mini()
    p = O.round_bf16(O.init_params(g, seed=1234))
    batch = O.synthetic_batch(g, 4, seed=1234)
    gold = np.load(os.path.join(GOLD, "mini_frozen.npz"))
    # the fixture and the regenerated inputs must be the same data
    assert np.array_equal(gold["language"], batch["language"].numpy())
    assert np.allclose(gold["vision_probe"], batch["vision"][:, 0, 0, :4, :4].numpy())
    m = VLBLitModule(_cfg())
    m.configure_model(state_dict=p, head_state=p)
    return m, p, batch, gold, g


def test_stage_parity_vs_golden(mini_frozen):
    """Every stage boundary of the forward against the committed fp32 goldens (bf16 activations)."""
    m, p, batch, gold, g = mini_frozen
    stages = {}
    vis = batch["vision"].to(m.device)
    ids = batch["language"].long().to(m.device)
    hidden, key_mask = m.backbone.forward(vis, ids, stages)
    B = 4
    assert rel_err(stages["vit_tokens"].view(B, g.num_frames, -1, g.vit_dim), torch.from_numpy(gold["vit_tokens"])) < 2e-2
    assert rel_err(stages["video_tokens"].view(B, g.vis_tokens, g.dim), torch.from_numpy(gold["video_tokens"])) < 3e-2
    assert np.array_equal(key_mask.bool().cpu().numpy(), gold["key_mask"])
    emb = stages["inputs_embeds"].view(B, g.max_len, g.dim)[:, ::7, ::5]
    assert rel_err(emb, torch.from_numpy(gold["inputs_embeds_probe"])) < 3e-2
    valid = torch.from_numpy(gold["key_mask"])[..., None]
    l0 = stages["layer_outputs"][0].view(B, g.max_len, g.dim).float().cpu()
    assert float(((l0 - torch.from_numpy(gold["layer0"])).abs() * valid).max() / np.abs(gold["layer0"]).max()) < 3e-2
    hid = hidden.view(B, g.max_len, g.dim).float().cpu()
    assert float(((hid - torch.from_numpy(gold["hidden"])).abs() * valid).max() / np.abs(gold["hidden"]).max()) < 3e-2
    assert torch.isfinite(hid).all()          # padded rows are don't-care but must stay finite (0 * NaN)


def test_training_step_loss_pred_grads(mini_frozen):
    """Predicted BOLD, loss (1e-3 rel, the north-star tolerance) and head gradients."""
    m, p, batch, gold, g = mini_frozen
    m.configure_optimizers()
    loss = m.training_step(batch)
    assert abs(float(loss) - float(gold["loss"])) / float(gold["loss"]) < 1e-3
    assert rel_err(m.head.pred, torch.from_numpy(gold["pred"])) < 3e-2
    assert rel_err(m.head.z, torch.from_numpy(gold["head_ln2"])) < 3e-2
    for n in ("ridge_layer.linear.weight", "ridge_layer.linear.bias", "layer_norm2.weight", "layer_norm2.bias",
              "layer_norm1.weight", "layer_norm1.bias"):
        assert rel_err(m.head.master[n].grad, torch.from_numpy(gold["grad::" + n])) < 4e-2, n


def test_optimizer_step_matches_oracle(mini_frozen):
    """clip(1.0) + AdamW + cosine LR on the head against the oracle's closed forms."""
    import vlb_oracle as O
    m, p, batch, gold, g = mini_frozen
    opt, sch = m.configure_optimizers()
    opt = opt[0]
    m.training_step(batch)
    names = [n for n, _ in m.trainable_named_parameters()]
    before = {n: m.head.master[n].clone().cpu() for n in names}
    grads = {n: m.head.master[n].grad.clone().cpu() for n in names}
    opt.step()
    sch[0]["scheduler"].step()
    clipped, total = O.clip_grad_norm(grads, 1.0)
    assert abs(opt.grad_norm() - float(total)) / float(total) < 1e-4
    for n in names:
        ref, _, _ = O.adamw_step(before[n], clipped[n], torch.zeros_like(before[n]), torch.zeros_like(before[n]), 1, 1e-4)
        assert (m.head.master[n].cpu() - ref).abs().max() < 2e-6, n
        assert torch.equal(m.head.compute[n].cpu(), m.head.master[n].cpu().to(torch.bfloat16))
    assert abs(opt.param_groups[0]["lr"] - O.cosine_lr(1e-4, 1, 50000)) < 1e-12


def test_live_oracle_other_seed(dev):
    """Fresh seed, live oracle on the host CPU (no fixture): loss within 1e-3 relative."""
    import vlb_oracle as O
    from phantom_vlb_amd.litmodule import VLBLitModule
    g = O.geometry_mini()
    p = O.round_bf16(O.init_params(g, seed=77))
    batch = O.synthetic_batch(g, 3, seed=78)
    with torch.no_grad():
        loss_ref, pred_ref = O.training_loss(p, batch, g)
    m = VLBLitModule(_cfg())
    m.configure_model(state_dict=p, head_state=p)
    m.configure_optimizers()
    loss = m.training_step(batch)
    assert abs(float(loss) - float(loss_ref)) / float(loss_ref) < 1e-3
    assert rel_err(m.head.pred, pred_ref) < 3e-2


def test_validation_step_contract(mini_frozen):
    m, p, batch, gold, g = mini_frozen
    out = m.validation_step(batch)
    assert set(out) == {"loss", "brain_preds", "brain_vals"}
    assert out["brain_preds"].shape == (4, 128) and out["brain_vals"].shape == (4, 128)
    assert "val/brain_loss" in m.logged


def test_vision_prefetch_is_bit_identical_and_consumed_once(dev):
    """Backbone.prefetch_video_tokens / video_tokens: the frozen vision side computed one step ahead on the side stream gives
    the very same bits as the in-line computation, is consumed exactly once, and a tensor that was modified in place since the
    prefetch (version counter) is recomputed in line.  Whole training steps with and without the prefetch give the same loss."""
    import dataclasses
    import vlb_oracle as O
    from phantom_vlb_amd.litmodule import VLBLitModule, VLBLitModuleConfig
    g = O.geometry_mini()
    p = O.round_bf16(O.init_params(g, seed=1234))
    batch = {k: (v.to(dev) if torch.is_tensor(v) and k not in ("language", "padvals") else v) for k, v in O.synthetic_batch(g, 4, seed=3).items()}
    cfg = VLBLitModuleConfig(model_path="none", freeze_backbone=True, use_lora=False, lora_r=None, lora_alpha=None, lora_dropout=None,
                             dropout_rate=0.0, num_target=128, l2_lambda=1e-3, lr=1e-4, betas=[0.9, 0.999], eps=1e-8, weight_decay=1e-2,
                             lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000, geometry="mini")
    m = VLBLitModule(cfg)
    m.configure_model(state_dict=p, head_state=p)
    m.configure_optimizers()
    bb = m.backbone
    inline = bb.video_tokens(batch["vision"]).clone()
    bb.prefetch_video_tokens(batch["vision"])
    assert len(bb._vis_queue) == 1
    got = bb.video_tokens(batch["vision"])
    assert len(bb._vis_queue) == 0 and torch.equal(got, inline)
    bb.prefetch_video_tokens(batch["vision"])
    batch["vision"].mul_(1.0)                               # in-place touch: the queued entry no longer describes this tensor
    again = bb.video_tokens(batch["vision"])
    assert len(bb._vis_queue) == 0 and torch.equal(again, inline)     # the stale entry is dropped at the lookup, result computed in line
    l0 = float(m.training_step(batch))
    m.prefetch_vision(batch)                                # deferred; its own step comes first -> computed in line, once
    assert len(bb._vis_pending) == 1 and not bb._vis_queue
    l1 = float(m.training_step(batch))
    assert not bb._vis_pending and not bb._vis_queue
    m.prefetch_vision(batch)                                # the bench / Trainer pattern: register batch i+1, run step i
    m.prefetch_vision(batch)
    l2 = float(m.training_step(batch))                      # consumes the first registration in line, launches the second behind its forward
    assert not bb._vis_pending and len(bb._vis_queue) == 1
    l3 = float(m.training_step(batch))                      # picks the prefetched features up
    torch.cuda.synchronize()
    assert l0 == l1 == l2 == l3 and len(bb._vis_queue) == 0
