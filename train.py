"""Drop-in for the reference's train.py (train.py:1-70): same CLI
``python train.py experiment=<name> subject=<sub-XX>`` and the same YAML keys.

Runs with real Hydra + Lightning when both are importable; otherwise uses the built-in config
loader and fit loop (phantom_vlb_amd.config / phantom_vlb_amd.trainer).  Comet logging is optional:
without ``comet_ml`` or credentials only the CSV logger is attached.
"""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def train(config: dict) -> None:
    import torch
    from phantom_vlb_amd.config import instantiate
    from phantom_vlb_amd.trainer import TrainableCheckpoint
    from src import LogValAccuracyCallback

    seed = int(config.get("random_state", 1234))
    torch.manual_seed(seed)                      # L.seed_everything(config.random_state), train.py:18
    import numpy as np
    import random
    np.random.seed(seed)
    random.seed(seed)

    callbacks = [
        TrainableCheckpoint(monitor="val/brain_loss", filename="best_brainloss", mode="min",
                            dirpath=config["output_dir"], save_last=True),
        LogValAccuracyCallback(),
    ]
    loggers = []
    comet_cfg = config.get("comet_logger")
    if comet_cfg and not any(u.startswith("my_") for u in config.get("_unresolved", [])):
        try:
            loggers.append(instantiate(comet_cfg))
        except Exception as e:            # no comet_ml / no network: CSV only
            print(f"[train] comet logger unavailable ({type(e).__name__}); continuing with CSV only")
    loggers.append(instantiate(config["cvs_logger"]))

    trainer = instantiate(config["trainer"], logger=loggers, callbacks=callbacks)
    datamodule = instantiate(config["datamodule"])
    print("[train] datasets:", datamodule.datasets.dset_names)
    litmodule = instantiate(config["litmodule"])
    trainer.fit(model=litmodule, datamodule=datamodule)
    callbacks[0].save(litmodule, os.path.join(config["output_dir"], "final.ckpt"), trainer.global_step)


if __name__ == "__main__":
    from phantom_vlb_amd.config import load_config
    cfg = load_config(os.path.join(ROOT, "config"), sys.argv[1:])
    train(cfg)
