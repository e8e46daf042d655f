"""Minimal fit loop standing in for ``lightning.pytorch.Trainer`` when Lightning is not installed.

Honours the keys the reference's YAML passes (config/experiment/*.yaml:42-51): ``precision``
(must be a bf16 mode - the kernels are bf16), ``gradient_clip_val`` (fused into the AdamW kernel),
``devices``/``num_nodes`` (one process per GPU via torchrun), ``max_epochs``, ``max_steps``,
``val_check_interval`` (fraction of an epoch), ``log_every_n_steps``; callbacks with the Lightning
hook names used by the reference (train.py:20-30); a CSV metrics file like CSVLogger's.
"""
from __future__ import annotations

import csv
import os
import time

import torch


class CSVLogger:
    def __init__(self, save_dir=".", name="run", **kw):
        self.dir = os.path.join(save_dir, name)
        os.makedirs(self.dir, exist_ok=True)
        self.path = os.path.join(self.dir, "metrics.csv")
        self.rows = []

    def log_hyperparams(self, params):
        pass

    def log_metrics(self, metrics: dict, step: int):
        self.rows.append({"step": step, **{k: float(v) for k, v in metrics.items()}})
        keys = sorted({k for r in self.rows for k in r})
        with open(self.path, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=keys)
            w.writeheader()
            w.writerows(self.rows)


class TrainableCheckpoint:
    """ModelCheckpoint(monitor="val/brain_loss", mode="min", save_last=True) for the TRAINABLE tensors only
    (head + LoRA) - the reference saves the whole frozen 7B each time and notes the TODO (train.py:21-27,60)."""

    def __init__(self, dirpath, monitor="val/brain_loss", mode="min", save_last=True, filename="best_brainloss", **kw):
        self.dirpath, self.monitor, self.mode, self.save_last, self.filename = dirpath, monitor, mode, save_last, filename
        self.best = None

    def save(self, module, path, step):
        if int(os.environ.get("RANK", "0")) != 0:
            return                      # data parallel: the trainables are replicated, rank 0 writes the one file
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        sd = {n: p.detach().cpu() for n, p in module.trainable_named_parameters()}
        opt = getattr(module, "optimizer", None)
        state = {"state_dict": sd, "global_step": step}
        if opt is not None:
            state.update(exp_avg=[m.cpu() for m in opt.m], exp_avg_sq=[v.cpu() for v in opt.v], opt_step=opt.step_count,
                         lr=opt.param_groups[0]["lr"])
        torch.save(state, path)

    def on_validation_end(self, trainer, module, metrics):
        val = metrics.get(self.monitor)
        if val is None:
            return
        val = float(val)
        better = self.best is None or (val < self.best if self.mode == "min" else val > self.best)
        if better:
            self.best = val
            self.save(module, os.path.join(self.dirpath, f"{self.filename}.ckpt"), trainer.global_step)
        if self.save_last:
            self.save(module, os.path.join(self.dirpath, "last.ckpt"), trainer.global_step)


def load_trainable_checkpoint(module, path):
    """Resume: restore trainables, Adam moments and the step counter."""
    st = torch.load(path, map_location="cpu")
    for n, p in module.trainable_named_parameters():
        p.copy_(st["state_dict"][n].to(p.device))
    opt = getattr(module, "optimizer", None)
    if opt is not None and "exp_avg" in st:
        for m, s in zip(opt.m, st["exp_avg"]):
            m.copy_(s.to(m.device))
        for v, s in zip(opt.v, st["exp_avg_sq"]):
            v.copy_(s.to(v.device))
        opt.step_count = st["opt_step"]
    for n in module.head.master:
        module.head.compute[n].copy_(module.head.master[n])
    if module.lora is not None:
        module.lora.refresh(from_master=True)
    return st.get("global_step", 0)


class Trainer:
    def __init__(self, precision="bf16-mixed", accelerator="gpu", gradient_clip_val=1.0, devices=1, num_nodes=1,
                 max_epochs=1, max_steps=-1, val_check_interval=1.0, log_every_n_steps=50, logger=None, callbacks=None,
                 limit_val_batches=None, **kw):
        if "bf16" not in str(precision):
            raise ValueError(f"precision={precision!r}: the libvlb kernels compute in bf16 (reference: bf16-mixed)")
        self.gradient_clip_val = float(gradient_clip_val or 0.0)
        self.max_epochs, self.max_steps = max_epochs, max_steps
        self.val_check_interval, self.log_every_n_steps = val_check_interval, log_every_n_steps
        self.loggers = [lg for lg in (logger if isinstance(logger, (list, tuple)) else [logger]) if lg is not None]
        self.callbacks = list(callbacks or [])
        self.limit_val_batches = limit_val_batches
        self.global_step = 0
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        want = devices if isinstance(devices, int) else (len(devices) if isinstance(devices, (list, tuple)) else 1)
        if want * int(num_nodes or 1) != self.world and self.rank == 0:
            import warnings
            warnings.warn(f"trainer.devices={devices} x num_nodes={num_nodes} but this job has WORLD_SIZE={self.world}: "
                          "one process drives one GPU here - launch with `torchrun --nproc-per-node N train.py ...` "
                          "to use N GPUs (clip-sharded data parallel over RCCL)")

    def _cb(self, hook, *a, **k):
        for c in self.callbacks:
            fn = getattr(c, hook, None)
            if fn is not None:
                fn(self, *a, **k)

    def _log(self, metrics):
        if self.rank == 0:
            for lg in self.loggers:
                lg.log_metrics(metrics, self.global_step)

    def validate(self, model, loader):
        self._cb("on_validation_epoch_start", model)
        tot, n = None, 0                     # summed on the device: one host sync per validation epoch, not per batch
        for bi, batch in enumerate(loader):
            if self.limit_val_batches is not None and bi >= self.limit_val_batches:
                break
            out = model.validation_step(batch)
            self._cb("on_validation_batch_end", model, out, batch, bi)
            tot = out["loss"].detach().double() if tot is None else tot + out["loss"].detach().double()
            n += 1
        self._cb("on_validation_epoch_end", model)
        metrics = {"val/brain_loss": (float(tot) if tot is not None else 0.0) / max(n, 1)}
        metrics.update({k: float(v) for k, v in getattr(model, "logged", {}).items() if k.startswith("val_corr_avg")})
        for c in self.callbacks:
            if hasattr(c, "on_validation_end") and isinstance(c, TrainableCheckpoint):
                c.on_validation_end(self, model, metrics)
        self._log(metrics)
        return metrics

    def fit(self, model, datamodule=None, ckpt_path=None):
        model.configure_model()
        model.config.gradient_clip_val = self.gradient_clip_val
        opts, scheds = model.configure_optimizers()
        opt, sched = opts[0], scheds[0]["scheduler"]
        if self.world > 1:
            from .parallel import attach_data_parallel, init_distributed, sync_module_states
            init_distributed()
            attach_data_parallel(model, opt)
            sync_module_states(model)
        if ckpt_path:
            self.global_step = load_trainable_checkpoint(model, ckpt_path)
        try:
            train_loader = datamodule.train_dataloader(rank=self.rank, world=self.world)
        except TypeError:
            train_loader = datamodule.train_dataloader()
        val_loader = datamodule.val_dataloader()
        if torch.cuda.is_available():        # overlap the next batch's host->device copy with the current step
            from .datamodule import DevicePrefetcher
            train_loader = DevicePrefetcher(train_loader, model.device)
            val_loader = DevicePrefetcher(val_loader, model.device)
        n_batches = len(train_loader)
        val_every = max(1, int(n_batches * self.val_check_interval)) if self.val_check_interval <= 1 else int(self.val_check_interval)
        t0 = time.time()
        for epoch in range(self.max_epochs):
            if hasattr(train_loader.sampler, "set_epoch"):
                train_loader.sampler.set_epoch(epoch)
            for bi, batch in enumerate(train_loader):
                loss = model.training_step(batch)
                opt.step()
                sched.step()
                self.global_step += 1
                if self.global_step % self.log_every_n_steps == 0:
                    self._log({"train/brain_loss": float(loss), "lr": opt.param_groups[0]["lr"], "epoch": epoch,
                               "elapsed_s": time.time() - t0})
                if (bi + 1) % val_every == 0:
                    self.validate(model, val_loader)
                if 0 < self.max_steps <= self.global_step:
                    return
        return

    def save_checkpoint(self, path):
        pass
