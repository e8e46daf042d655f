"""Host-side boundary checks that need no GPU: Lightning `_target_`s go to the real Lightning when it is importable (the
module / datamodule / callback classes then subclass its bases) and to the built-in runner otherwise, `find_all_linear_names` restates the reference's walk (litmodule :36-55), the
LR monitor / checkpoint callbacks the reference's train.py wires (train.py:20-30,58) exist under the built-in
Trainer."""
import os
import sys
import textwrap

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_with_fake_lightning(tmp_path, code, env_extra=None):
    """Run `code` in a fresh interpreter that finds tests/fake_lightning's package as `lightning` (a clean process: the
    package decides its base classes when it is imported)."""
    import subprocess
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fake_lightning
    root = fake_lightning.write(tmp_path / "site")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([root, ROOT, os.path.join(ROOT, "oracle")]))
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


def test_with_lightning_installed_the_yaml_targets_and_base_classes_are_lightnings(tmp_path):
    """Reference train.py:41-56 instantiates `lightning.pytorch.Trainer` from the YAML and hands it VLBLitModule /
    VLBDataModule / LogValAccuracyCallback.  With a `lightning` package importable the three classes subclass Lightning's
    bases (Trainer.fit type-checks them) and VLB_TRAINER=lightning instantiates those targets as written; the default
    (and VLB_TRAINER=builtin) is the built-in runner - the Lightning bridge is opt-in."""
    code = """
        import os, sys
        import lightning.pytorch as lp
        from lightning.pytorch.callbacks import Callback
        from phantom_vlb_amd import config as C, trainer as T
        from src.litmodule import VLBLitModule
        from src.datamodule import VLBDataModule
        from src import LogValAccuracyCallback
        assert issubclass(VLBLitModule, lp.LightningModule) and issubclass(VLBDataModule, lp.LightningDataModule)
        assert issubclass(LogValAccuracyCallback, Callback)
        cfg = C.load_config("config", ["experiment=VLB_vllama2_friends_lora", "subject=sub-01", "output_dir=/tmp/x"])
        assert cfg["trainer"]["_target_"] == "lightning.pytorch.Trainer"          # the reference's YAML, unchanged
        tr = C.instantiate(cfg["trainer"], logger=[], callbacks=[])
        want_builtin = os.environ.get("VLB_TRAINER", "builtin") == "builtin"
        assert (type(tr) is T.Trainer) == want_builtin and (type(tr) is lp.Trainer) == (not want_builtin)
        assert tr.gradient_clip_val == 1
        lg = C.instantiate(cfg["cvs_logger"])
        assert (type(lg) is T.CSVLogger) == want_builtin
        print("OK", type(tr).__module__)
    """
    assert "OK lightning.pytorch" in _run_with_fake_lightning(tmp_path, code, {"VLB_TRAINER": "lightning"})
    assert "OK phantom_vlb_amd.trainer" in _run_with_fake_lightning(tmp_path, code, {"VLB_TRAINER": "builtin"})
    assert "OK phantom_vlb_amd.trainer" in _run_with_fake_lightning(tmp_path, code)      # the default is the built-in runner


def test_without_lightning_the_builtin_runner_serves_the_yaml(tmp_path):
    from phantom_vlb_amd import config as C, trainer as T
    assert C.use_builtin_trainer()              # this environment has no Lightning
    cfg = C.load_config(os.path.join(ROOT, "config"), ["experiment=VLB_vllama2_friends_lora", "subject=sub-01", f"output_dir={tmp_path}"])
    tr = C.instantiate(cfg["trainer"], logger=[], callbacks=[])
    assert type(tr) is T.Trainer and tr.gradient_clip_val == 1.0 and tr.max_epochs == cfg["trainer"]["max_epochs"]
    assert type(C.instantiate(cfg["cvs_logger"])) is T.CSVLogger
    assert type(C.instantiate({"_target_": "lightning.pytorch.callbacks.LearningRateMonitor", "logging_interval": "epoch"})) \
        is T.LearningRateMonitor


def test_train_py_wires_the_reference_callbacks():
    src = open(os.path.join(ROOT, "train.py")).read()
    for needle in ("TrainableCheckpoint(monitor=\"val/brain_loss\"", "LearningRateMonitor(logging_interval=\"epoch\")",
                   "LogValAccuracyCallback()", "trainer.save_checkpoint(config[\"output_dir\"])", "trainer.fit("):
        assert needle in src, needle


def test_find_all_linear_names_on_a_torch_module_tree():
    """Same walk as the reference: nn.Linear leaves outside mm_projector / vision_tower / vision_resampler, minus lm_head."""
    from phantom_vlb_amd.litmodule import find_all_linear_names
    import torch.nn as nn

    class Attn(nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj, self.k_proj, self.v_proj, self.o_proj = (nn.Linear(4, 4) for _ in range(4))

    class Mlp(nn.Module):
        def __init__(self):
            super().__init__()
            self.gate_proj, self.up_proj, self.down_proj = nn.Linear(4, 8), nn.Linear(4, 8), nn.Linear(8, 4)
            self.act = nn.SiLU()

    class Layer(nn.Module):
        def __init__(self):
            super().__init__()
            self.self_attn, self.mlp, self.norm = Attn(), Mlp(), nn.LayerNorm(4)

    class Inner(nn.Module):
        def __init__(self):
            super().__init__()
            self.layers = nn.ModuleList([Layer(), Layer()])
            self.embed_tokens = nn.Embedding(10, 4)
            self.vision_tower = nn.Sequential(nn.Linear(4, 4))          # skipped: multimodal keyword
            self.mm_projector = nn.ModuleDict({"readout": nn.Linear(4, 4), "conv": nn.Conv2d(4, 4, 1)})

    class Model(nn.Module):
        def __init__(self):
            super().__init__()
            self.model, self.lm_head = Inner(), nn.Linear(4, 10)
    names = find_all_linear_names(Model())
    assert names == sorted(["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"])
    assert find_all_linear_names(nn.Linear(2, 2)) == [""]          # like the reference: a root-level Linear has the empty name


def test_litmodule_keeps_the_reference_hook_surface():
    """Reference src/litmodule/videollama2_vlb_litmodule.py:160-379: the hooks Trainer.fit and train.py reach for, by name."""
    import inspect
    from src.litmodule import VLBLitModule, VLBLitModuleConfig
    for name in ("configure_model", "make_weight_mask", "forward", "training_step", "validation_step", "configure_optimizers",
                 "log", "parameters", "state_dict", "configure_gradient_clipping", "transfer_batch_to_device", "prefetch_vision"):
        assert callable(getattr(VLBLitModule, name, None)), name
    assert list(inspect.signature(VLBLitModule.training_step).parameters) == ["self", "batch"]        # no batch_idx (reference :259)
    assert list(inspect.signature(VLBLitModule.validation_step).parameters) == ["self", "batch"]
    fields = [f for f in VLBLitModuleConfig.__dataclass_fields__][:16]
    assert fields == ["model_path", "freeze_backbone", "use_lora", "lora_r", "lora_alpha", "lora_dropout", "dropout_rate", "num_target",
                      "l2_lambda", "lr", "betas", "eps", "weight_decay", "lr_scheduler_name", "last_epoch", "t_max"]
