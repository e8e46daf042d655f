"""Runner-level checks on the GPU: train.py on the mini synthetic experiment, trainables-only checkpoint
+ resume, and the fsdp.yaml-equivalent sharded layer store giving identical results (world = 1)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(**kw):
    from phantom_vlb_amd.litmodule import VLBLitModuleConfig
    base = dict(model_path="none", freeze_backbone=True, use_lora=False, lora_r=None, lora_alpha=None, lora_dropout=None,
                dropout_rate=0.0, num_target=128, l2_lambda=1e-3, lr=1e-3, betas=[0.9, 0.999], eps=1e-8,
                weight_decay=1e-2, lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000, geometry="mini")
    base.update(kw)
    return VLBLitModuleConfig(**base)


def test_train_py_mini_experiment(tmp_path):
    """python train.py experiment=VLB_mini_synthetic subject=sub-99 : loss goes down, CSV + checkpoints written."""
    out = tmp_path / "run"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "experiment=VLB_mini_synthetic", "subject=sub-99",
                        f"output_dir={out}", "trainer.max_epochs=6"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    import csv
    rows = list(csv.DictReader(open(out / "mini_sub-99" / "metrics.csv")))
    train = [float(x["train/brain_loss"]) for x in rows if x.get("train/brain_loss")]
    val = [float(x["val/brain_loss"]) for x in rows if x.get("val/brain_loss")]
    assert len(train) >= 6 and len(val) >= 2
    assert train[-1] < train[0]                      # the head learns on the fixed synthetic clips
    for f in ("last.ckpt", "final.ckpt"):
        assert (out / f).exists()
    # reference train.py:24 filename="best_brainloss_{epoch}-{step}" as Lightning formats it; save_top_k=1 keeps ONE best
    import re
    best = [f for f in os.listdir(out) if f.startswith("best_brainloss_")]
    assert len(best) == 1 and re.fullmatch(r"best_brainloss_epoch=\d+-step=\d+\.ckpt", best[0]), best
    st = torch.load(out / "final.ckpt", map_location="cpu")
    assert set(st["state_dict"]) == {"layer_norm1.weight", "layer_norm1.bias", "layer_norm2.weight", "layer_norm2.bias",
                                     "ridge_layer.linear.weight", "ridge_layer.linear.bias"}      # trainables only


def _dm(batch_size=2):
    from phantom_vlb_amd.datamodule import VLBDataModule, VLBDataModuleConfig
    return VLBDataModule(VLBDataModuleConfig(lazyload_path="synthetic:3x4", subject="sub-01", seasons=["s1"], delay=3, window=3,
                                             random_state=1234, shuffle_val_data=False, batch_size=batch_size, geometry="mini",
                                             num_target=128))


def test_checkpoint_resume_through_trainer_fit_is_exact(dev, tmp_path):
    """6 steps in one run  ==  4 steps, checkpoint, and `Trainer.fit(ckpt_path=...)` in a fresh process-state for
    steps 5-6 - with LoRA dropout 0.1 AND head dropout 0.1 active: the LR schedule, the counter-based dropout
    seeds, the Adam moments and the clip order all continue (bit-exact masters and bf16 copies)."""
    from phantom_vlb_amd.litmodule import VLBLitModule
    from phantom_vlb_amd.trainer import TrainableCheckpoint, Trainer
    kw = dict(use_lora=True, freeze_backbone=False, lora_r=16, lora_alpha=32, lora_dropout=0.1, dropout_rate=0.1)
    a = VLBLitModule(_cfg(**kw))
    ta = Trainer(max_epochs=2, max_steps=6, val_check_interval=1.0, log_every_n_steps=1,
                 callbacks=[TrainableCheckpoint(str(tmp_path), filename="best")])
    ta.fit(a, _dm())
    assert ta.global_step == 6
    st = torch.load(tmp_path / "last.ckpt", map_location="cpu", weights_only=False)
    assert st["global_step"] == 4 and st["lr_scheduler"]["last_epoch"] == 4 and st["rng"] == {"head_step": 4, "lora_step": 4}
    assert st["state_dict"]["model.layers.0.self_attn.q_proj.lora_B.weight"].shape == (a.geometry.heads * a.geometry.head_dim, 16)
    b = VLBLitModule(_cfg(**kw))
    tb = Trainer(max_epochs=2, max_steps=6, val_check_interval=1.0, log_every_n_steps=1)
    tb.fit(b, _dm(), ckpt_path=str(tmp_path / "last.ckpt"))
    assert tb.global_step == 6 and b.optimizer.step_count == 6
    assert b.optimizer.param_groups[0]["lr"] == a.optimizer.param_groups[0]["lr"]
    assert torch.equal(a.flat.master, b.flat.master)
    assert torch.equal(a.flat.compute, b.flat.compute)
    # trainer.save_checkpoint(config.output_dir) (reference train.py:58): a directory gets final.ckpt
    out = tb.save_checkpoint(str(tmp_path))
    assert out.endswith("final.ckpt") and os.path.exists(out)


def test_trainable_state_dict_round_trips_through_configure_model(dev):
    """checkpoint `state_dict` (peft layout) merged into the backbone's state dict -> a fresh configure_model
    reproduces masters, bf16 copies and predictions; rank 8 < 16 keeps its [8,in] / [out,8] shapes."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vlb_oracle as O
    from phantom_vlb_amd.litmodule import VLBLitModule
    from phantom_vlb_amd.synthetic import synthetic_batch
    g = O.geometry_mini()
    p = O.round_bf16(O.init_params(g, seed=5))
    for r in (16, 8):
        kw = dict(use_lora=True, freeze_backbone=False, lora_r=r, lora_alpha=32, lora_dropout=0.0)
        a = VLBLitModule(_cfg(**kw))
        a.configure_model(state_dict=p, head_state=p)
        opt, _ = a.configure_optimizers()
        batch = synthetic_batch(a.geometry, 2, seed=5, device=a.device)
        for _ in range(2):
            a.training_step(batch); opt[0].step()
        sd = a.trainable_state_dict()
        assert sd["model.layers.1.mlp.down_proj.lora_A.weight"].shape == (r, g.ff)
        assert sd["model.layers.1.mlp.down_proj.lora_B.weight"].shape == (g.dim, r)
        assert float(sd["model.layers.1.mlp.down_proj.lora_B.weight"].abs().max()) > 0          # B has trained away from 0
        b = VLBLitModule(_cfg(**kw))
        b.configure_model(state_dict={**p, **sd})
        b.configure_optimizers()
        assert torch.equal(a.flat.master, b.flat.master) and torch.equal(a.flat.compute, b.flat.compute)
        assert torch.equal(a.validation_step(batch)["brain_preds"], b.validation_step(batch)["brain_preds"])
        if r < 16:            # padded adapter rows never leave zero
            assert float(a.lora.master["model.layers.0.self_attn.q_proj.lora_A.weight"][r:].abs().max()) == 0.0
            assert float(a.lora.master["model.layers.0.self_attn.q_proj.lora_B.weight"][r:].abs().max()) == 0.0


@pytest.mark.parametrize("lora", [False, True])
def test_sharded_layer_store_same_result(dev, lora):
    """Backbone.enable_sharding() (world 1: the gather is a copy on the side stream) changes nothing."""
    from phantom_vlb_amd.litmodule import VLBLitModule
    from phantom_vlb_amd.synthetic import synthetic_batch
    kw = dict(use_lora=True, freeze_backbone=False, lora_r=16, lora_alpha=32, lora_dropout=0.0) if lora else {}
    outs = []
    for shard in (False, True):
        m = VLBLitModule(_cfg(**kw))
        m.configure_model()
        if shard:
            m.backbone.enable_sharding()
            assert m.backbone.w.layers[0]["wqkv"] is None
        m.configure_optimizers()
        batch = synthetic_batch(m.geometry, 2, seed=9, device=m.device)
        loss = m.training_step(batch)
        torch.cuda.synchronize()
        outs.append((float(loss), m.flat.grad.clone()))
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1])


def test_upstream_named_safetensors_checkpoint_loads(dev, tmp_path):
    """model_path = a local directory of *.safetensors with UPSTREAM tensor names (what a real
    VideoLLaMA2 checkpoint looks like): same prediction as handing the state dict in directly."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vlb_oracle as O
    from safetensors.torch import save_file
    from phantom_vlb_amd.litmodule import VLBLitModule
    from phantom_vlb_amd.synthetic import synthetic_batch
    g = O.geometry_mini()
    p = O.round_bf16(O.init_params(g, seed=21))
    backbone = {k: v.to(torch.bfloat16).contiguous() for k, v in p.items() if k.startswith("model.")}
    keys = sorted(backbone)
    save_file({k: backbone[k] for k in keys[: len(keys) // 2]}, str(tmp_path / "model-00001-of-00002.safetensors"))
    save_file({k: backbone[k] for k in keys[len(keys) // 2:]}, str(tmp_path / "model-00002-of-00002.safetensors"))
    a = VLBLitModule(_cfg(model_path=str(tmp_path)))
    a.configure_model(head_state=p)
    b = VLBLitModule(_cfg())
    b.configure_model(state_dict=p, head_state=p)
    batch = synthetic_batch(a.geometry, 2, seed=3, device=a.device)
    pa = a.validation_step(batch)["brain_preds"]
    pb = b.validation_step(batch)["brain_preds"]
    assert torch.equal(pa, pb)


def test_device_prefetcher_hands_over_identical_batches(dev):
    """Side-stream H2D prefetch: same values as the plain loader, big tensors on the device, ids / padvals
    left on the host (the packed row layout is sized from them without a sync)."""
    from phantom_vlb_amd.datamodule import DevicePrefetcher, VLBDataModule, VLBDataModuleConfig
    dm = VLBDataModule(VLBDataModuleConfig(lazyload_path="synthetic:4x5", subject="sub-01", seasons=["s1"], delay=3,
                                           window=3, random_state=1234, shuffle_val_data=False, batch_size=2,
                                           geometry="mini", num_target=128))
    loader = dm.val_dataloader()
    plain = list(loader)
    pre = list(DevicePrefetcher(loader, dev))
    assert len(pre) == len(plain) > 1
    for a, b in zip(plain, pre):
        assert set(a) == set(b)
        for k in a:
            if k in ("language", "padvals"):
                assert b[k].device.type == "cpu"
            else:
                assert b[k].is_cuda
            assert torch.equal(a[k], b[k].cpu()), k
    assert len(DevicePrefetcher(loader, dev)) == len(loader)


def test_full_finetune_through_trainer_fit_and_exact_resume(dev, tmp_path):
    """BASELINE configs[4] (freeze_backbone=False, use_lora=False) through the runner: everything but the vision tower
    trains; the checkpoint carries the backbone store (masters + both moments) and `fit(ckpt_path=)` resumes bit-exactly."""
    from phantom_vlb_amd.litmodule import VLBLitModule
    from phantom_vlb_amd.trainer import TrainableCheckpoint, Trainer
    kw = dict(freeze_backbone=False, use_lora=False, dropout_rate=0.1)
    a = VLBLitModule(_cfg(**kw))
    ta = Trainer(max_epochs=2, max_steps=6, val_check_interval=1.0, log_every_n_steps=1,
                 callbacks=[TrainableCheckpoint(str(tmp_path), filename="best")])
    ta.fit(a, _dm())
    assert a.full is not None and ta.global_step == 6
    st = torch.load(tmp_path / "last.ckpt", map_location="cpu", weights_only=False)
    assert st["global_step"] == 4 and len(st["stores"]) == 1 and st["stores"][0]["master"].numel() == a.full.flat.numel
    assert "model.layers.0.self_attn.q_proj.weight" in st["state_dict"] and "model.mm_projector.sampler.0.weight" in st["state_dict"]
    assert not any(k.startswith("model.vision_tower") for k in st["state_dict"])          # the tower is frozen, not saved
    b = VLBLitModule(_cfg(**kw))
    tb = Trainer(max_epochs=2, max_steps=6, val_check_interval=1.0, log_every_n_steps=1)
    tb.fit(b, _dm(), ckpt_path=str(tmp_path / "last.ckpt"))
    assert torch.equal(a.full.flat.master, b.full.flat.master) and torch.equal(a.full.flat.compute, b.full.flat.compute)
    assert torch.equal(a.flat.master, b.flat.master)
    lw = b.backbone.w.layers[0]
    assert torch.equal(lw["wo_t"], lw["wo"].t())


def test_lightning_automatic_optimisation_drives_the_module_bit_for_bit(dev, tmp_path):
    """Reference train.py:41-56: `lightning.pytorch.Trainer(precision="bf16-mixed", gradient_clip_val=1).fit(litmodule,
    datamodule)`.  With a `lightning` package importable (tests/fake_lightning: its Trainer.fit runs Lightning's
    automatic-optimisation sequence training_step -> zero_grad -> loss.backward() -> configure_gradient_clipping ->
    optimizer.step -> scheduler.step) VLBLitModule is a LightningModule and survives that loop: 4 steps with LoRA dropout and
    head dropout on leave EXACTLY the parameters the built-in runner produces from the same seeds - and the loss handed to
    `loss.backward()` carries a graph although the kernels have none."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fake_lightning
    root = fake_lightning.write(tmp_path / "site")
    code = r'''
import torch
import lightning.pytorch as lp
from lightning.pytorch.callbacks import LearningRateMonitor
from src.litmodule import VLBLitModule, VLBLitModuleConfig
from src.datamodule import VLBDataModule, VLBDataModuleConfig
from src import LogValAccuracyCallback
from phantom_vlb_amd import trainer as T

def cfg():
    return VLBLitModuleConfig(model_path="none", freeze_backbone=False, use_lora=True, lora_r=16, lora_alpha=32, lora_dropout=0.1,
                              dropout_rate=0.1, num_target=128, l2_lambda=1e-3, lr=1e-3, betas=[0.9, 0.999], eps=1e-8, weight_decay=1e-2,
                              lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000, geometry="mini")

def dm():
    return VLBDataModule(VLBDataModuleConfig(lazyload_path="synthetic:3x4", subject="sub-01", seasons=["s1"], delay=3, window=3,
                                             random_state=1234, shuffle_val_data=False, batch_size=2, geometry="mini", num_target=128))

a = VLBLitModule(cfg())
assert isinstance(a, lp.LightningModule)
ta = lp.Trainer(precision="bf16-mixed", gradient_clip_val=1, max_epochs=2, max_steps=4, callbacks=[LogValAccuracyCallback(), LearningRateMonitor()])
ta.fit(model=a, datamodule=dm())
assert ta.global_step == 4 and a.optimizer.step_count == 4 and a.optimizer.max_norm == 1.0
assert "train/brain_loss" in ta.logged_metrics and all(l == l for l in ta.losses)
# Lightning's zero_grad ran between training_step and backward: the gradients were re-attached by loss.backward()
n, p = a.trainable_named_parameters()[-1]
assert isinstance(p, torch.nn.Parameter) and p.grad is not None and p.grad.data_ptr() == a.lora.grads[n].data_ptr()
loss = a.training_step(a.transfer_batch_to_device(next(iter(dm().train_dataloader())), a.device))
assert loss.requires_grad and loss.grad_fn is not None
sd = a.state_dict()
assert "ridge_layer.linear.weight" in sd and any(".lora_A." in k for k in sd) and not any("embed_tokens" in k for k in sd)

b = VLBLitModule(cfg())
tb = T.Trainer(precision="bf16-mixed", gradient_clip_val=1, max_epochs=2, max_steps=4, val_check_interval=1.0)
tb.fit(b, dm())
assert tb.global_step == 4
# 4 optimiser steps each; `a` ran one extra training_step (no optimiser step) after its fit
assert torch.equal(a.flat.master, b.flat.master) and torch.equal(a.flat.compute, b.flat.compute), "parameters differ"
assert a.optimizer.param_groups[0]["lr"] == b.optimizer.param_groups[0]["lr"]
print("BRIDGE_OK")
'''
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([root, ROOT, os.path.join(ROOT, "oracle")]))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0 and "BRIDGE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_lightning_checkpoint_resume_is_exact_and_multi_device_is_refused(dev, tmp_path):
    """on the Lightning path ModelCheckpoint stores `module.state_dict()` + `optimizer.state_dict()` +
    `on_save_checkpoint`.  VlbAdamW.state_dict carries the Adam moments and the bias-correction step, the module's hook the
    dropout counters: 2 steps + save + resume in a NEW module + 2 steps == 4 straight steps, bit for bit (LoRA + both
    dropouts on).  And a Lightning Trainer with more than one device is refused in on_fit_start (DDP would step on
    unreduced gradients: the kernels' gradients never flow through autograd)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fake_lightning
    root = fake_lightning.write(tmp_path / "site")
    code = r'''
import sys, torch
import lightning.pytorch as lp
from src.litmodule import VLBLitModule, VLBLitModuleConfig
from src.datamodule import VLBDataModule, VLBDataModuleConfig

def cfg():
    return VLBLitModuleConfig(model_path="none", freeze_backbone=False, use_lora=True, lora_r=16, lora_alpha=32, lora_dropout=0.1,
                              dropout_rate=0.1, num_target=128, l2_lambda=1e-3, lr=1e-3, betas=[0.9, 0.999], eps=1e-8, weight_decay=1e-2,
                              lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000, geometry="mini")

def dm():
    return VLBDataModule(VLBDataModuleConfig(lazyload_path="synthetic:3x4", subject="sub-01", seasons=["s1"], delay=3, window=3,
                                             random_state=1234, shuffle_val_data=False, batch_size=2, geometry="mini", num_target=128))

def tr(**kw):
    return lp.Trainer(precision="bf16-mixed", gradient_clip_val=1, max_epochs=3, **kw)

path = sys.argv[1]
a = VLBLitModule(cfg()); ta = tr(max_steps=4); ta.fit(model=a, datamodule=dm())
b = VLBLitModule(cfg()); tb = tr(max_steps=2); tb.fit(model=b, datamodule=dm()); tb.save_checkpoint(path)
ck = torch.load(path, map_location="cpu", weights_only=False)
v = ck["optimizer_states"][0]["vlb"]
assert v["step_count"] == 2 and float(v["stores"][0]["m"].abs().max()) > 0 and float(v["stores"][0]["v"].abs().max()) > 0
assert ck["vlb_rng"]["head_step"] == 2 and any(".lora_B." in k for k in ck["state_dict"])
c = VLBLitModule(cfg()); tc = tr(max_steps=4); tc.fit(model=c, datamodule=dm(), ckpt_path=path)
assert tc.global_step == 4 and c.optimizer.step_count == 4
assert torch.equal(a.flat.master, c.flat.master) and torch.equal(a.flat.compute, c.flat.compute), "resumed parameters differ"
assert torch.equal(a.flat.m, c.flat.m) and torch.equal(a.flat.v, c.flat.v), "resumed moments differ"
assert a.optimizer.param_groups[0]["lr"] == c.optimizer.param_groups[0]["lr"]
assert not torch.equal(a.flat.master, b.flat.master)
try:
    tr(max_steps=1, devices=2).fit(model=VLBLitModule(cfg()), datamodule=dm())
    raise SystemExit("a 2-device Lightning Trainer was accepted")
except ValueError as e:
    assert "ONE device" in str(e)
print("RESUME_OK")
'''
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([root, ROOT, os.path.join(ROOT, "oracle")]), VLB_TRAINER="lightning")
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path / "lightning.ckpt")], capture_output=True, text=True, timeout=900,
                       env=env, cwd=ROOT)
    assert r.returncode == 0 and "RESUME_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
