"""A miniature ``lightning.pytorch`` for tests (the real package is not installed in this environment): LightningModule /
LightningDataModule / Callback base classes with the attributes the reference's classes touch, and a Trainer whose ``fit``
performs the AUTOMATIC-OPTIMISATION sequence of Lightning 2.x for one optimiser -

    optimizer.step(closure)  with  closure = { training_step(batch) [under bf16 autocast] -> optimizer.zero_grad()
                                               -> loss.backward() -> configure_gradient_clipping(...) }
    then lr_scheduler.step()

- exactly the calls a module must survive to be driven by ``lightning.pytorch.Trainer(precision="bf16-mixed",
gradient_clip_val=1)`` as the reference's train.py:41-56 does - plus ``save_checkpoint`` / ``fit(ckpt_path=)`` in the order
Lightning's checkpoint connector uses (module hooks + state_dict before the optimisers exist, optimiser / scheduler
state_dicts after).  ``write(dir)`` lays the package down under ``dir``."""
import os
import textwrap

PYTORCH_INIT = '''
import torch


class LightningModule(torch.nn.Module):
    FAKE = True

    def __init__(self):
        super().__init__()
        self._trainer = None

    @property
    def trainer(self):
        if self._trainer is None:
            raise RuntimeError(f"{type(self).__name__} is not attached to a `Trainer`.")
        return self._trainer

    @trainer.setter
    def trainer(self, t):
        self._trainer = t

    def log(self, name, value, **kw):
        self.trainer.logged_metrics[name] = float(value)

    def configure_model(self):
        pass

    def on_fit_start(self):
        pass

    def on_save_checkpoint(self, checkpoint):
        pass

    def on_load_checkpoint(self, checkpoint):
        pass

    def transfer_batch_to_device(self, batch, device, dataloader_idx=0):
        return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}

    def configure_gradient_clipping(self, optimizer, gradient_clip_val=None, gradient_clip_algorithm=None):
        if gradient_clip_val:
            torch.nn.utils.clip_grad_norm_(self.parameters(), gradient_clip_val)


class LightningDataModule:
    FAKE = True

    def __init__(self):
        pass


class SingleDeviceStrategy:
    pass


class DDPStrategy:
    pass


class Trainer:
    FAKE = True

    def __init__(self, precision="32-true", gradient_clip_val=None, max_epochs=1, max_steps=-1, callbacks=None, logger=None,
                 accumulate_grad_batches=1, devices=1, strategy="auto", **kw):
        self.precision, self.gradient_clip_val = precision, gradient_clip_val
        self.max_epochs, self.max_steps, self.accumulate_grad_batches = max_epochs, max_steps, accumulate_grad_batches
        self.callbacks, self.kw = list(callbacks or []), kw
        self.global_step, self.logged_metrics, self.losses = 0, {}, []
        n = devices if isinstance(devices, int) else len(devices)
        self.world_size = n                                     # one process per device under Lightning's launchers
        self.strategy = DDPStrategy() if (n > 1 or strategy == "ddp") else SingleDeviceStrategy()

    def fit(self, model, datamodule=None, ckpt_path=None):
        from lightning.pytorch.callbacks import Callback
        assert isinstance(model, LightningModule), "Trainer.fit: `model` must be a LightningModule"
        assert isinstance(datamodule, LightningDataModule), "Trainer.fit: `datamodule` must be a LightningDataModule"
        assert all(isinstance(c, Callback) for c in self.callbacks), "callbacks must subclass Callback"
        self.model = model
        model.trainer = self
        model.configure_model()
        ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False) if ckpt_path else None
        if ckpt is not None:                 # Lightning 2.x: _restore_modules_and_callbacks BEFORE the optimisers exist
            model.on_load_checkpoint(ckpt)
            model.load_state_dict(ckpt["state_dict"])
        opts, scheds = model.configure_optimizers()
        opt, sched = opts[0], scheds[0]["scheduler"]
        assert scheds[0]["interval"] == "step" and isinstance(opt, torch.optim.Optimizer)
        self.optimizers, self.lr_schedulers = [opt], [sched]
        if ckpt is not None:                 # ... and restore_training_state after strategy.setup
            opt.load_state_dict(ckpt["optimizer_states"][0])
            sched.load_state_dict(ckpt["lr_schedulers"][0])
            self.global_step = ckpt["global_step"]
        model.on_fit_start()
        model.train()
        loader = datamodule.train_dataloader()
        start_epoch, skip = divmod(self.global_step, max(len(loader), 1)) if ckpt is not None else (0, 0)
        for epoch in range(start_epoch, self.max_epochs):
            if hasattr(loader.sampler, "set_epoch"):
                loader.sampler.set_epoch(epoch)
            for batch_idx, batch in enumerate(loader):
                if epoch == start_epoch and batch_idx < skip:
                    continue                                    # mid-epoch resume: Lightning's loop state does the same
                batch = model.transfer_batch_to_device(batch, model.device, 0)

                def closure():
                    with torch.autocast("cuda", dtype=torch.bfloat16, enabled="bf16" in str(self.precision)):
                        loss = model.training_step(batch)
                    opt.zero_grad()
                    loss.backward()                       # automatic optimisation: needs a graph
                    model.configure_gradient_clipping(opt, self.gradient_clip_val, None)
                    return loss
                self.losses.append(float(opt.step(closure=closure)))
                sched.step()
                self.global_step += 1
                if 0 < self.max_steps <= self.global_step:
                    return

    def save_checkpoint(self, filepath, weights_only=False):
        """What Lightning's checkpoint connector dumps for one optimiser (dump_checkpoint): the module's state_dict, the
        optimiser's and the scheduler's state_dicts, the step - after giving the module its on_save_checkpoint hook."""
        ckpt = {"state_dict": self.model.state_dict(), "global_step": self.global_step,
                "optimizer_states": [o.state_dict() for o in self.optimizers],
                "lr_schedulers": [s.state_dict() for s in self.lr_schedulers]}
        self.model.on_save_checkpoint(ckpt)
        torch.save(ckpt, filepath)


def seed_everything(seed):
    torch.manual_seed(seed)
'''

CALLBACKS_INIT = '''
class Callback:
    FAKE = True


class ModelCheckpoint(Callback):
    def __init__(self, **kw):
        self.kw = kw


class LearningRateMonitor(Callback):
    def __init__(self, **kw):
        self.kw = kw
'''

LOGGERS_INIT = '''
class CSVLogger:
    FAKE = True

    def __init__(self, **kw):
        self.kw = kw
'''


def write(root):
    pkg = os.path.join(str(root), "lightning")
    for sub in ("pytorch/callbacks", "pytorch/loggers"):
        os.makedirs(os.path.join(pkg, sub), exist_ok=True)
    for rel, text in (("__init__.py", ""), ("pytorch/__init__.py", PYTORCH_INIT), ("pytorch/callbacks/__init__.py", CALLBACKS_INIT),
                      ("pytorch/loggers/__init__.py", LOGGERS_INIT)):
        with open(os.path.join(pkg, rel), "w") as f:
            f.write(textwrap.dedent(text))
    return str(root)
