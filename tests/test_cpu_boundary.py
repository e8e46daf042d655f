"""Host-side boundary checks that need no GPU: Lightning `_target_`s are routed to the built-in runner even when a
`lightning` package is importable, `find_all_linear_names` restates the reference's walk (litmodule :36-55), the
LR monitor / checkpoint callbacks the reference's train.py wires (train.py:20-30,58) exist under the built-in
Trainer."""
import os
import sys
import textwrap

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def fake_lightning(tmp_path, monkeypatch):
    """A minimal importable `lightning.pytorch` (Trainer, LightningModule, callbacks, loggers) on sys.path, whose
    Trainer.fit does what the real one does in automatic optimisation: call loss.backward()."""
    pkg = tmp_path / "lightning"
    (pkg / "pytorch" / "callbacks").mkdir(parents=True)
    (pkg / "pytorch" / "loggers").mkdir(parents=True)
    (pkg / "__init__.py").write_text("")
    (pkg / "pytorch" / "__init__.py").write_text(textwrap.dedent('''
        class LightningModule:
            FAKE = True
        class LightningDataModule:
            FAKE = True
        class Trainer:
            FAKE = True
            def __init__(self, **kw):
                self.kw = kw
            def fit(self, model, datamodule=None):
                loss = model.training_step(next(iter(datamodule.train_dataloader())))
                loss.backward()          # automatic optimisation: needs an autograd graph
        def seed_everything(s):
            pass
    '''))
    (pkg / "pytorch" / "callbacks" / "__init__.py").write_text(textwrap.dedent('''
        class Callback:
            FAKE = True
        class ModelCheckpoint(Callback):
            def __init__(self, **kw): self.kw = kw
        class LearningRateMonitor(Callback):
            def __init__(self, **kw): self.kw = kw
    '''))
    (pkg / "pytorch" / "loggers" / "__init__.py").write_text(textwrap.dedent('''
        class CSVLogger:
            FAKE = True
            def __init__(self, **kw): self.kw = kw
    '''))
    monkeypatch.syspath_prepend(str(tmp_path))
    for k in [k for k in sys.modules if k == "lightning" or k.startswith("lightning.")]:
        monkeypatch.delitem(sys.modules, k)
    import lightning.pytorch as lp
    assert lp.Trainer.FAKE
    yield lp
    for k in [k for k in sys.modules if k == "lightning" or k.startswith("lightning.")]:
        sys.modules.pop(k, None)


def test_lightning_targets_route_to_the_builtin_runner_even_when_lightning_is_importable(fake_lightning, tmp_path):
    import importlib
    from phantom_vlb_amd import config as C, trainer as T
    cfg = C.load_config(os.path.join(ROOT, "config"), ["experiment=VLB_vllama2_friends_lora", "subject=sub-01",
                                                       f"output_dir={tmp_path}"])
    assert cfg["trainer"]["_target_"] == "lightning.pytorch.Trainer"          # the reference's YAML, unchanged
    tr = C.instantiate(cfg["trainer"], logger=[], callbacks=[])
    assert type(tr) is T.Trainer and not hasattr(tr, "FAKE")
    assert tr.gradient_clip_val == 1.0 and tr.max_epochs == cfg["trainer"]["max_epochs"]
    lg = C.instantiate(cfg["cvs_logger"])
    assert type(lg) is T.CSVLogger
    assert type(C.instantiate({"_target_": "lightning.pytorch.callbacks.LearningRateMonitor", "logging_interval": "epoch"})) \
        is T.LearningRateMonitor
    # the module / datamodule / callback classes never take the Lightning base classes, importable or not
    for mod in ("phantom_vlb_amd.litmodule", "phantom_vlb_amd.datamodule", "phantom_vlb_amd.utils"):
        m = importlib.reload(importlib.import_module(mod))
        for name in ("VLBLitModule", "VLBDataModule", "LogValAccuracyCallback"):
            cls = getattr(m, name, None)
            if cls is not None:
                assert not any(getattr(b, "FAKE", False) for b in cls.__mro__), (mod, name)


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
