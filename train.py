"""Drop-in for the reference's train.py (train.py:1-70): same CLI
``python train.py experiment=<name> subject=<sub-XX>`` and the same YAML keys.

The YAML is read by the built-in Hydra-subset loader (phantom_vlb_amd.config).  ``_target_: lightning.pytorch.Trainer`` and
the callbacks are served by the built-in fit loop (phantom_vlb_amd.trainer), which honours the same keys -
the default, and the loop behind every measured number and the multi-GPU path.  VLB_TRAINER=lightning opts in to the real
Lightning objects (VLBLitModule / VLBDataModule / LogValAccuracyCallback subclass the Lightning base classes whenever the
package is importable, so the reference's own unchanged train.py can drive them too): ONE device, exercised only against
tests/fake_lightning.py so far.  Comet logging is optional: without ``comet_ml`` or
credentials only the CSV logger is attached.
"""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def train(config: dict) -> None:
    import torch
    from phantom_vlb_amd.config import instantiate, use_builtin_trainer
    from src import LogValAccuracyCallback
    if use_builtin_trainer():
        from phantom_vlb_amd.trainer import LearningRateMonitor, TrainableCheckpoint
    else:                                        # reference train.py:3-4,20-30: the real callbacks
        from lightning.pytorch.callbacks import LearningRateMonitor, ModelCheckpoint as TrainableCheckpoint

    seed = int(config.get("random_state", 1234))
    torch.manual_seed(seed)                      # L.seed_everything(config.random_state), train.py:18
    import numpy as np
    import random
    np.random.seed(seed)
    random.seed(seed)

    callbacks = [
        TrainableCheckpoint(monitor="val/brain_loss", filename="best_brainloss_{epoch}-{step}", mode="min",     # train.py:21-27
                            dirpath=config["output_dir"], save_last=True),
        LearningRateMonitor(logging_interval="epoch"),
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
    trainer.save_checkpoint(config["output_dir"])          # reference train.py:58 -> <output_dir>/final.ckpt


if __name__ == "__main__":
    from phantom_vlb_amd.config import load_config
    cfg = load_config(os.path.join(ROOT, "config"), sys.argv[1:])
    train(cfg)
