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


class LearningRateMonitor:
    """``lightning.pytorch.callbacks.LearningRateMonitor(logging_interval=...)`` (reference train.py:28): logs the
    optimiser's learning rate under Lightning's key ``lr-AdamW`` every epoch (or every logged step)."""

    def __init__(self, logging_interval="epoch", **kw):
        self.logging_interval = logging_interval

    def on_train_epoch_start(self, trainer, module):
        if self.logging_interval != "step":
            trainer._log({"lr-AdamW": module.optimizer.param_groups[0]["lr"]})

    def on_train_batch_start(self, trainer, module, batch, batch_idx):
        if self.logging_interval == "step" and trainer.global_step % trainer.log_every_n_steps == 0:
            trainer._log({"lr-AdamW": module.optimizer.param_groups[0]["lr"]})


def trainable_state(module, step: int, to_host: bool = True) -> dict | None:
    """Everything a bit-exact resume needs, trainables only (head + LoRA / trained backbone tensors), gathered from
    the shards under data parallelism (a collective: call on every rank).  ``state_dict`` uses the upstream / peft
    layouts (``lora_B.weight`` is [out, r]) so it can be handed to ``configure_model(state_dict=...)`` or to peft.
    ``to_host=False`` (every rank but the writer): take part in the gathers, build nothing, return None - the host
    copies of a 7B full fine-tune's master and moments are 84 GB and only rank 0 writes the file."""
    opt = getattr(module, "optimizer", None)
    shs = [sh for sh in (getattr(module, "sharded", None), getattr(module, "sharded_backbone", None)) if sh is not None]
    for sh in shs:
        sh.gather_masters()                # the optimiser updates the owned slices only: refresh the full-size masters
    n_stores = len(opt.flats) if opt is not None else 0

    def gathered(name, i=0):
        staged = opt.flats[i].master if name == "master" and opt.shardeds[i] is not None else None
        t = staged if staged is not None else opt.full_state(name, i)          # collective under data parallelism
        return t.cpu() if to_host else None

    state = {}
    try:
        if opt is not None:
            state.update(exp_avg=gathered("m"), exp_avg_sq=gathered("v"))
            # further flat stores (full fine-tune: the backbone in kernel layouts) travel whole: master + both moments
            state["stores"] = [{k: gathered(k, i) for k in ("master", "m", "v")} for i in range(1, n_stores)]
        if not to_host:
            return None
        state.update(state_dict=module.trainable_state_dict(), global_step=step, format=2)
    finally:
        for sh in shs:
            sh.release_staging()           # FULL_SHARD keeps no standing full-size master
    if opt is not None:
        state.update(opt_step=opt.step_count, flat_offsets={n: (o, k) for n, (o, k, _) in module.flat.offsets.items()},
                     lr=opt.param_groups[0]["lr"])
    sch = getattr(module, "scheduler", None)
    if sch is not None:
        state["lr_scheduler"] = sch.state_dict()
    state["rng"] = module.rng_state()
    return state


class TrainableCheckpoint:
    """ModelCheckpoint(monitor="val/brain_loss", mode="min", save_last=True) for the TRAINABLE tensors only
    (head + LoRA) - the reference saves the whole frozen 7B each time and notes the TODO (train.py:21-27,60)."""

    def __init__(self, dirpath, monitor="val/brain_loss", mode="min", save_last=True, filename="best_brainloss_{epoch}-{step}",
                 save_top_k=1, **kw):
        self.dirpath, self.monitor, self.mode, self.save_last, self.filename = dirpath, monitor, mode, save_last, filename
        self.save_top_k = save_top_k
        self.best, self.best_model_path = None, ""

    def format_checkpoint_name(self, epoch: int, step: int) -> str:
        """Lightning's naming (ModelCheckpoint.format_checkpoint_name with auto_insert_metric_name, its default): the
        reference's ``filename="best_brainloss_{epoch}-{step}"`` (train.py:24) becomes ``best_brainloss_epoch=3-step=120.ckpt``."""
        name = self.filename.replace("{epoch}", f"epoch={epoch}").replace("{step}", f"step={step}")
        return os.path.join(self.dirpath, f"{name}.ckpt")

    def save(self, module, path, step):
        writer = int(os.environ.get("RANK", "0")) == 0
        state = trainable_state(module, step, to_host=writer)       # gathers the shards: every rank takes part
        if not writer:
            return                      # data parallel: rank 0 writes the one file
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        torch.save(state, path)

    def on_validation_end(self, trainer, module, metrics):
        val = metrics.get(self.monitor)
        if val is None:
            return
        val = float(val)
        better = self.best is None or (val < self.best if self.mode == "min" else val > self.best)
        if better:
            self.best = val
            path = self.format_checkpoint_name(getattr(trainer, "current_epoch", 0), trainer.global_step)
            self.save(module, path, trainer.global_step)
            old, self.best_model_path = self.best_model_path, path
            if self.save_top_k == 1 and old and old != path and os.path.exists(old) and int(os.environ.get("RANK", "0")) == 0:
                os.remove(old)          # save_top_k=1 (Lightning's default): the previous best goes
        if self.save_last:
            self.save(module, os.path.join(self.dirpath, "last.ckpt"), trainer.global_step)


def load_trainable_checkpoint(module, path):
    """Resume: trainables, Adam moments and step count, LR-scheduler state, and the dropout counters / generator
    (LoRA's counter-based seed, the head's mask generator) - so the resumed run replays neither LR nor masks."""
    st = torch.load(path, map_location="cpu", weights_only=False)
    module.load_trainable_state_dict(st["state_dict"])
    opt = getattr(module, "optimizer", None)
    if opt is not None and "exp_avg" in st:
        opt.load_full_state("master", module.flat.master)
        opt.load_full_state("m", st["exp_avg"])
        opt.load_full_state("v", st["exp_avg_sq"])
        opt.step_count = st["opt_step"]
        for i, store in enumerate(st.get("stores", []), start=1):
            for k in ("master", "m", "v"):
                opt.load_full_state(k, store[k], i)
            opt.compute_from_master(i)
        for fn in opt.post_step:
            fn()                          # derived layouts (LoRA A^T / B pads, W^T copies) from the restored weights
    sch = getattr(module, "scheduler", None)
    if sch is not None and "lr_scheduler" in st:
        sch.load_state_dict(st["lr_scheduler"])
        if opt is not None:
            opt.param_groups[0]["lr"] = st["lr"]
    if "rng" in st:
        module.set_rng_state(st["rng"])
    return st.get("global_step", 0)


def _iter_selected(loader, select=None, limit=None):
    """(index, batch) of the batches with select(index) true, stopping in front of index ``limit``.  A DevicePrefetcher
    stages only those (datamodule.DevicePrefetcher.iter_selected); a plain loader is simply filtered."""
    if hasattr(loader, "iter_selected"):
        return loader.iter_selected(select, limit)

    def gen():
        for i, b in enumerate(loader):
            if limit is not None and i >= limit:
                return
            if select is None or select(i):
                yield i, b
    return gen()


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
        self.global_step, self.current_epoch = 0, 0
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
        """Validation epoch.  Data parallel: ranks take rank-strided batches (the frozen weights are replicated, so
        a validation forward needs no collective), and the loss sum / batch count / Pearson sums are all-reduced
        once at the end; with sharded frozen weights (per-layer all-gathers in every forward) all ranks keep
        running every batch so their collective sequences stay aligned."""
        self._cb("on_validation_epoch_start", model)
        tot, n = None, 0                     # summed on the device: one host sync per validation epoch, not per batch
        strided = self.world > 1 and getattr(model.backbone, "store", None) is None
        mine = (lambda bi: bi % self.world == self.rank) if strided else None
        for bi, batch in _iter_selected(loader, mine, self.limit_val_batches):
            out = model.validation_step(batch)
            self._cb("on_validation_batch_end", model, out, batch, bi)
            tot = out["loss"].detach().double() if tot is None else tot + out["loss"].detach().double()
            n += 1
        if strided:
            import torch.distributed as dist
            t = torch.zeros(2, dtype=torch.float64, device=model.device)
            if tot is not None:
                t[0] = tot
            t[1] = n
            dist.all_reduce(t)
            tot, n = t[0], int(t[1].item())
            for c in self.callbacks:
                if hasattr(c, "all_reduce_sums"):
                    c.all_reduce_sums(model)         # every rank takes part, also one that drew no batch
        self._cb("on_validation_epoch_end", model)
        metrics = {"val/brain_loss": (float(tot) if tot is not None else 0.0) / max(n, 1)}
        metrics.update({k: float(v) for k, v in getattr(model, "logged", {}).items() if k.startswith("val_corr_avg")})
        for c in self.callbacks:
            if hasattr(c, "on_validation_end") and isinstance(c, TrainableCheckpoint):
                c.on_validation_end(self, model, metrics)
        self._log(metrics)
        return metrics

    def fit(self, model, datamodule=None, ckpt_path=None):
        self.model = model
        model.trainer = self
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
            # ... and start the frozen vision side of batch i+1 on a side stream under step i (VLBLitModule.prefetch_vision)
            hook = getattr(model, "prefetch_vision", None)
            drop = getattr(model, "discard_prefetched_vision", None)
            train_loader = DevicePrefetcher(train_loader, model.device, on_staged=hook, on_discard=drop)
            val_loader = DevicePrefetcher(val_loader, model.device, on_staged=hook, on_discard=drop)
        n_batches = len(train_loader)
        val_every = max(1, int(n_batches * self.val_check_interval)) if self.val_check_interval <= 1 else int(self.val_check_interval)
        t0 = time.time()
        start_epoch, skip = divmod(self.global_step, max(n_batches, 1)) if ckpt_path else (0, 0)
        for epoch in range(start_epoch, self.max_epochs):
            self.current_epoch = epoch
            if hasattr(train_loader.sampler, "set_epoch"):
                train_loader.sampler.set_epoch(epoch)
            self._cb("on_train_epoch_start", model)
            seen = (lambda bi: bi >= skip) if epoch == start_epoch and skip else None     # resumed mid-epoch: the batches
            batches = _iter_selected(train_loader, seen, None)                            # before the checkpoint are not staged
            try:
                for bi, batch in batches:
                    self._cb("on_train_batch_start", model, batch, bi)
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
            finally:
                batches.close()              # a batch staged ahead but never consumed is dropped (DevicePrefetcher.on_discard)
        return

    def save_checkpoint(self, filepath, weights_only: bool = False):
        """``trainer.save_checkpoint(config.output_dir)`` (reference train.py:58).  The reference passes the output
        DIRECTORY; Lightning would try to write a file of that name - here a directory gets ``final.ckpt`` inside."""
        model = getattr(self, "model", None)
        if model is None:
            raise RuntimeError("save_checkpoint() before fit(): no module attached")
        if os.path.isdir(filepath) or not os.path.splitext(filepath)[1]:
            filepath = os.path.join(filepath, "final.ckpt")
        TrainableCheckpoint(os.path.dirname(filepath)).save(model, filepath, self.global_step)
        return filepath
