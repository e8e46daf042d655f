"""Just enough of Hydra/OmegaConf to run the reference's YAML unchanged when they are not installed.

Supports what ``train.py`` + ``config/`` use: a ``defaults`` list with
``experiment: <name>`` selected on the command line (``experiment=VLB_vllama2_friends_lora``),
``# @package _global_`` experiment files merged at the root, ``key.sub=value`` overrides,
``${key}`` interpolation, and ``_target_`` instantiation of nested configs.
"""
from __future__ import annotations

import importlib
import os
import re

import yaml

# Lightning `_target_`s of the reference's YAML.  By default they go to the built-in runner (phantom_vlb_amd.trainer), which
# honours the same keys and is the loop every measurement and the data-parallel path use.  VLB_TRAINER=lightning opts in to
# the real `lightning.pytorch` objects (VLBLitModule is a LightningModule whenever Lightning is importable and bridges its
# explicit backward to the automatic-optimisation loop, litmodule._ExplicitLoss; single device only, checked in
# on_fit_start).  The bridge has only ever run against tests/fake_lightning.py - the real package is not installed here.
ROUTED_TARGETS = {
    "lightning.pytorch.Trainer": "phantom_vlb_amd.trainer.Trainer",
    "lightning.pytorch.loggers.CSVLogger": "phantom_vlb_amd.trainer.CSVLogger",
    "lightning.pytorch.callbacks.LearningRateMonitor": "phantom_vlb_amd.trainer.LearningRateMonitor",
    "lightning.pytorch.callbacks.ModelCheckpoint": "phantom_vlb_amd.trainer.TrainableCheckpoint",
}


def use_builtin_trainer() -> bool:
    """True when the YAML's Lightning targets are served by phantom_vlb_amd.trainer: always, unless VLB_TRAINER=lightning
    asks for the real package (and then its absence is an ImportError, not a silent fallback)."""
    want = os.environ.get("VLB_TRAINER", "builtin")
    if want in ("builtin", "auto", ""):
        return True
    if want != "lightning":
        raise ValueError(f"VLB_TRAINER={want!r}: expected 'builtin' or 'lightning'")
    import lightning.pytorch  # noqa: F401
    return False


def _merge(a: dict, b: dict) -> dict:
    out = dict(a)
    for k, v in b.items():
        out[k] = _merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
    return out


def _set(cfg: dict, dotted: str, value):
    keys = dotted.split(".")
    d = cfg
    for k in keys[:-1]:
        d = d.setdefault(k, {})
    d[keys[-1]] = yaml.safe_load(value) if isinstance(value, str) else value


def _lookup(cfg, dotted):
    d = cfg
    for k in dotted.split("."):
        d = d[k]
    return d


def _resolve(node, root, missing):
    if isinstance(node, dict):
        return {k: _resolve(v, root, missing) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root, missing) for v in node]
    if isinstance(node, str) and re.fullmatch(r"[+-]?\d+(\.\d*)?[eE][+-]?\d+", node):
        return float(node)                     # OmegaConf reads 1e-4 as a float; YAML 1.1 (PyYAML) does not
    if isinstance(node, str):
        full = re.fullmatch(r"\$\{([^}]+)\}", node)
        if full:
            try:
                return _resolve(_lookup(root, full.group(1)), root, missing)
            except KeyError:
                missing.add(full.group(1))
                return None

        def sub(m):
            try:
                return str(_resolve(_lookup(root, m.group(1)), root, missing))
            except KeyError:
                missing.add(m.group(1))
                return ""
        return re.sub(r"\$\{([^}]+)\}", sub, node)
    return node


def load_config(config_dir: str, overrides: list[str], config_name: str = "base") -> dict:
    with open(os.path.join(config_dir, f"{config_name}.yaml")) as f:
        cfg = yaml.safe_load(f) or {}
    groups = {}
    for item in cfg.pop("defaults", []) or []:
        if isinstance(item, dict):
            groups.update(item)
    plain = []
    for ov in overrides:
        k, _, v = ov.partition("=")
        if k in groups or os.path.isdir(os.path.join(config_dir, k)):
            groups[k] = v
        else:
            plain.append((k, v))
    for grp, name in groups.items():
        if not name or name == "null":
            continue
        path = os.path.join(config_dir, grp, f"{name}.yaml")
        if not os.path.exists(path):
            continue                              # e.g. logger/comet.yaml is git-ignored upstream
        with open(path) as f:
            text = f.read()
        sub = yaml.safe_load(text) or {}
        cfg = _merge(cfg, sub) if "@package _global_" in text else _merge(cfg, {grp: sub})
    for k, v in plain:
        _set(cfg, k, v)
    missing = set()
    cfg = _resolve(cfg, cfg, missing)
    cfg["_unresolved"] = sorted(missing)
    return cfg


def _locate(path: str):
    mod, _, name = path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def instantiate(node, **extra):
    """hydra.utils.instantiate for dict configs (recursive on nested ``_target_`` values)."""
    if not isinstance(node, dict) or "_target_" not in node:
        return node
    kwargs = {k: instantiate(v) for k, v in node.items() if k != "_target_"}
    kwargs.update(extra)
    target = str(node["_target_"]).strip()
    if target in ROUTED_TARGETS and use_builtin_trainer():
        target = ROUTED_TARGETS[target]
    cls = _locate(target)
    return cls(**kwargs)
