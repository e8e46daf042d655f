"""Drop-in mirror of the reference's ``src.litmodule`` (VLBLitModule / VLBLitModuleConfig).

Same class names, the same 16 config fields, the same hooks with the same meaning
(src/litmodule/videollama2_vlb_litmodule.py:126-379): ``configure_model``, ``make_weight_mask``,
``forward(x_video, x_lang, weight_mask, attention_mask)`` -> ``(regression_output, l2_reg)``,
``training_step(batch)`` -> scalar loss (no batch_idx), ``validation_step(batch)`` ->
``{'loss','brain_preds','brain_vals'}``, ``configure_optimizers()`` -> ``([opt],[{scheduler...}])``,
``self.log("train/brain_loss", ...)``.  Module attribute names ``nnmodule``, ``hrf_layer``,
``ridge_layer``, ``layer_norm1``, ``layer_norm2``, ``dropout`` are kept.

What differs, on purpose: all arithmetic runs in libvlb HIP kernels; the backward pass is explicit
(``training_step`` leaves fp32 gradients in ``param.grad`` of the trainable masters and returns the
loss), because there is no autograd graph through hand-written kernels.
"""
from __future__ import annotations

import os
import warnings
from dataclasses import dataclass

import torch

from . import ops
from .backbone import Backbone, Weights
from .geometry import Geometry, geometry_7b, geometry_mini
from .head import HEAD_PARAMS, BrainHead
from .optim import VlbAdamW


def _lightning():
    """``lightning.pytorch`` when it is importable, else None (this container and the GPU boxes: absent)."""
    try:
        import lightning.pytorch as lp
        return lp
    except Exception:
        return None


_LP = _lightning()

if _LP is not None:
    class _Base(_LP.LightningModule):
        """Real Lightning is installed: VLBLitModule IS a ``LightningModule`` (reference litmodule :160), so the reference's
        unchanged ``train.py:41-56`` can hand it to ``lightning.pytorch.Trainer.fit``.  The kernels have no autograd graph;
        the bridge to Lightning's automatic optimisation is ``_ExplicitLoss`` below (``loss.backward()`` hands out the
        gradients ``training_step`` already computed) plus the hooks ``configure_gradient_clipping`` (the YAML's
        ``gradient_clip_val: 1`` is applied inside the fused AdamW kernel), ``transfer_batch_to_device`` and
        ``state_dict`` on VLBLitModule."""

        def __init__(self):
            super().__init__()
            self.logged: dict[str, float] = {}

        def log(self, name, value, **kw):
            self.logged[name] = value
            tr = getattr(self, "_trainer", None)
            if tr is not None and isinstance(tr, _LP.Trainer):       # the built-in runner reads self.logged instead
                super().log(name, value, **kw)
else:
    class _Base:
        """The slice of ``lightning.pytorch.LightningModule`` the reference's module uses (``self.log``, ``self.device``,
        train/eval mode, ``self.trainer``) for environments without Lightning; ``train.py`` then runs the built-in
        ``phantom_vlb_amd.trainer.Trainer`` (same YAML keys).  With Lightning installed the class above is used instead."""

        def __init__(self):
            self.training = True
            self.logged: dict[str, float] = {}
            self.trainer = None

        def log(self, name, value, **kw):
            self.logged[name] = value

        def train(self, mode: bool = True):
            self.training = mode
            return self

        def eval(self):
            return self.train(False)


class _ExplicitLoss(torch.autograd.Function):
    """The scalar ``training_step`` returns: the loss value the kernels computed, attached to the trainable Parameters so that
    Lightning's automatic optimisation can call ``loss.backward()``.  The backward pass ran explicitly inside
    ``training_step`` (hand-written kernels, no autograd tape); ``backward`` here only (re-)attaches those gradient buffers as
    the Parameters' ``.grad`` - Lightning zeroes the gradients between ``training_step`` and ``backward`` - and returns no
    autograd gradients, so nothing is accumulated twice.  The upstream gradient must be 1 (no gradient accumulation /
    loss scaling: the reference uses neither; ``on_fit_start`` checks)."""

    @staticmethod
    def forward(ctx, loss, module, *params):
        ctx.module = module
        ctx.n = len(params)
        return loss.detach().clone()

    @staticmethod
    def backward(ctx, gout):
        ctx.module._attach_gradients()
        return (None, None) + (None,) * ctx.n


class _LinearInfo:
    """Stand-in for an ``nn.Linear`` leaf when a weight store lists its module tree (``Backbone.named_modules``)."""
    is_linear = True

    def __init__(self, out_features, in_features):
        self.out_features, self.in_features = out_features, in_features


def find_all_linear_names(model):
    """reference :36-55 (from VideoLLaMA2's trainer): leaf names of every linear layer outside the multimodal
    modules, minus ``lm_head`` - the LoRA ``target_modules``.  ``model`` is anything with ``named_modules()``:
    a torch module tree (``nn.Linear`` leaves) or this package's ``Backbone`` (leaves listed from its weights)."""
    lora_module_names = set()
    multimodal_keywords = ["mm_projector", "vision_tower", "vision_resampler"]
    for name, module in model.named_modules():
        if any(k in name for k in multimodal_keywords):
            continue
        if isinstance(module, torch.nn.Linear) or getattr(module, "is_linear", False):
            names = name.split(".")
            lora_module_names.add(names[0] if len(names) == 1 else names[-1])
    lora_module_names.discard("lm_head")
    return sorted(lora_module_names)


@dataclass
class VLBLitModuleConfig:
    """Field for field the reference's dataclass (src/litmodule/...:138-153) plus optional extras."""
    model_path: str
    freeze_backbone: bool
    use_lora: bool
    lora_r: int | None
    lora_alpha: int | None
    lora_dropout: float | None
    dropout_rate: float
    num_target: int
    l2_lambda: float
    lr: float
    betas: list[float]
    eps: float
    weight_decay: float
    lr_scheduler_name: str
    last_epoch: int
    t_max: int
    # ---- extras (not in the reference; all optional)
    geometry: str = "7b"            # "7b" | "mini"
    init_seed: int = 1234
    gradient_clip_val: float = 1.0  # the Trainer's gradient_clip_val, applied inside the fused AdamW
    pack_tokens: bool = True        # drop each clip's padded tail rows (flash-attn varlen equivalent)
    fp8_gemm: bool = False          # full fine-tune only: decoder forward / dgrad GEMMs on the MX-fp8 MFMA path (configs[4])

    def __post_init__(self):
        self.dtype = torch.bfloat16      # reference :155
        self.device_map = "auto"         # reference :157 (degenerate on one device)


def resolve_geometry(cfg: VLBLitModuleConfig) -> Geometry:
    kw = dict(num_target=cfg.num_target, l2_lambda=cfg.l2_lambda)
    if cfg.use_lora:
        kw.update(lora_r=cfg.lora_r, lora_alpha=cfg.lora_alpha)
    return geometry_mini(**kw) if cfg.geometry == "mini" else geometry_7b(**kw)


class _Attr:
    """Tiny namespace so ``self.nnmodule.config.hidden_size`` etc. read as in the reference."""
    def __init__(self, **kw):
        self.__dict__.update(kw)


class VLBLitModule(_Base):
    def __init__(self, config: VLBLitModuleConfig) -> None:
        super().__init__()
        self.config = config
        from .parallel import local_device_index
        self._device = torch.device("cuda", local_device_index())
        self._step = 0
        self.world_size = 1
        self.rank = 0
        self.pack_tokens = bool(getattr(config, "pack_tokens", True))

    @property
    def device(self):
        return self._device

    # ------------------------------------------------------------------ model
    def configure_model(self, state_dict: dict | None = None, head_state: dict | None = None) -> None:
        """reference :206-226.  The hub checkpoint is unreachable offline: weights are
        random-initialised on the device unless a state dict (upstream naming) is handed in.  Head tensors
        (``layer_norm1.weight`` ...) and adapters (peft layout: ``...lora_A.weight`` [r,in], ``...lora_B.weight``
        [out,r]) are taken from ``head_state`` or, when present there, from ``state_dict`` itself - so a
        ``trainable_state_dict()`` / checkpoint ``state_dict`` merged into the backbone's loads back unchanged."""
        if getattr(self, "nnmodule", None) is not None:
            return
        cfg = self.config
        # reference :86-111: freeze_backbone=False and use_lora=False -> everything but the vision tower trains
        full_ft = not cfg.freeze_backbone and not cfg.use_lora
        g = self.geometry = resolve_geometry(cfg)
        dev = self.device
        torch.cuda.set_device(dev)
        if state_dict is None:
            if cfg.model_path and os.path.isdir(cfg.model_path):
                state_dict = load_safetensors_dir(cfg.model_path)
            else:
                warnings.warn(f"model_path={cfg.model_path!r} is not a local directory: random-initialising "
                              f"the {cfg.geometry} architecture (no network / HF cache in this environment)")
                state_dict = Weights.random_state_dict(g, dev, seed=cfg.init_seed)
        weights = Weights(g, state_dict, dev, keep_transposed=bool(cfg.use_lora) or full_ft, gate_up_interleaved=not full_ft)
        lora_state = {k: v for k, v in state_dict.items() if ".lora_" in k} or None
        if head_state is None and all(n in state_dict for n in HEAD_PARAMS):
            head_state = {n: state_dict[n] for n in HEAD_PARAMS}
        del state_dict
        self.backbone = Backbone(g, weights)
        self.nnmodule = _Attr(config=_Attr(hidden_size=g.dim, tokenizer_model_max_length=g.max_len,
                                           num_frames=g.num_frames, use_cache=False), backbone=self.backbone)
        self.head = BrainHead(g.dim, cfg.num_target, cfg.l2_lambda, g.ln_eps, dev, sd=head_state, seed=cfg.init_seed)
        # reference attribute names
        self.hrf_layer = self.head
        self.ridge_layer = self.head
        self.layer_norm1 = self.head
        self.layer_norm2 = self.head
        self.dropout = _Attr(p=cfg.dropout_rate)
        self.lora = self.full = None
        if full_ft:
            from .fullft import FullFineTune
            self.full = FullFineTune(g, self.backbone, dev, fp8_gemm=bool(getattr(cfg, "fp8_gemm", False)))
        if cfg.use_lora:
            from .lora import LoraState
            lora_sd = None if head_state is None and lora_state is None else {**(head_state or {}), **(lora_state or {})}
            self.lora = LoraState(g, weights, cfg.lora_r, cfg.lora_alpha, cfg.lora_dropout or 0.0, dev, seed=cfg.init_seed,
                                  sd=lora_sd, target_modules=find_all_linear_names(self.backbone))

    def _named_masters(self):
        """(name, fp32 master tensor) of everything AdamW updates: head always; LoRA A/B when use_lora
        (reference :86-120: backbone frozen / peft freezes the base)."""
        out = [(n, self.head.master[n]) for n in HEAD_PARAMS]
        if self.lora is not None:
            out += self.lora.named_masters()
        if self.full is not None:       # full fine-tune: every backbone tensor outside the vision tower (kernel layouts)
            f = self.full.flat
            if f.master is None:        # FULL_SHARD (parallel.attach_data_parallel): no full-size master; the handles are placeholders
                cache = self.__dict__.get("_param_cache", {})
                out += [(f"backbone.{n}", cache[f"backbone.{n}"].data) for n in f.offsets if f"backbone.{n}" in cache]
            else:
                out += [(f"backbone.{n}", f.view(f.master, n)) for n in f.offsets]
        return out

    def trainable_named_parameters(self):
        """(name, ``nn.Parameter``) of the trainables.  Each Parameter ALIASES its fp32 master (the flat buffer the fused
        AdamW updates in place): the objects are stable across calls, so an optimiser's ``param_groups``, Lightning's
        ``zero_grad`` / ``loss.backward()`` and ``.grad`` all refer to the same things; they are re-made only when a master
        moved (``configure_optimizers`` re-points the masters into the flat store)."""
        cache = self.__dict__.setdefault("_param_cache", {})
        out = []
        for n, t in self._named_masters():
            hit = cache.get(n)
            if hit is None or hit.data_ptr() != t.data_ptr() or hit.shape != t.shape:
                hit = cache[n] = torch.nn.Parameter(t, requires_grad=True)
            out.append((n, hit))
        return out

    def parameters(self, recurse: bool = True):
        return [p for _, p in self.trainable_named_parameters()]

    def _attach_gradients(self):
        """``.grad`` of every trainable Parameter <- the gradient buffer the explicit backward filled (aliases, no copies)."""
        masters = dict(self._named_masters())
        for n, p in self.trainable_named_parameters():
            if n.startswith("backbone."):
                continue        # bf16 gradients (like the reference's bf16 parameters'): read them with self.full.flat.g_(name)
            g = self.head.grads[n] if n in self.head.grads else self.lora.grads[n]
            p.grad = g
            masters[n].grad = g             # the plain master views (head.master[n], lora.master[n]) carry it too

    def trainable_state_dict(self) -> dict:
        """Trainables on the host under their upstream / peft names and layouts (LoRA B as [out, r], rank padding
        removed): what checkpoints store and what ``configure_model(state_dict=...)`` / peft accept."""
        sd = {n: self.head.master[n].detach().cpu().clone() for n in HEAD_PARAMS}
        if self.lora is not None:
            sd.update({n: t.cpu() for n, t in self.lora.state_dict().items()})
        if self.full is not None:
            sd.update(self.full.state_dict())
        return sd

    def load_trainable_state_dict(self, sd: dict) -> None:
        """Inverse of ``trainable_state_dict`` into the fp32 masters; bf16 copies and derived layouts are rebuilt."""
        for n in HEAD_PARAMS:
            self.head.master[n].copy_(sd[n].to(self.device, torch.float32))
            self.head.compute[n].copy_(self.head.master[n])
        if self.lora is not None:
            self.lora.load_state_dict(sd)
        # full fine-tune: the backbone masters travel as the flat store itself (trainer.trainable_state 'stores')

    # ------------------------------------------------------------------ dropout randomness (counter based)
    def _dropout_seed(self) -> int:
        """Head dropout seed of the current step: (init_seed, rank, step) - ranks draw different masks for their
        clips and a resumed run continues the sequence from the restored step counter."""
        x = (self.config.init_seed * 0x9E3779B1 + (self.rank + 1) * 0x7F4A7C15 + self._step * 0x85EBCA6B + 0x165667B1) & 0xFFFFFFFF
        return x or 1

    def rng_state(self) -> dict:
        return {"head_step": self._step, "lora_step": None if self.lora is None else self.lora.step}

    def set_rng_state(self, st: dict) -> None:
        self._step = int(st.get("head_step", 0))
        if self.lora is not None and st.get("lora_step") is not None:
            self.lora.step = int(st["lora_step"])

    # ------------------------------------------------------------------ pieces of the step
    def make_weight_mask(self, pad_vals, vis_weights, lang_weights, lang_len, max_len):
        """reference :178-203 - one launch; values rounded to bf16 like the reference's mask."""
        g = self.geometry
        feature_len = vis_weights.shape[1] * g.ds_grid * g.ds_grid + lang_len - 1
        assert feature_len == max_len
        dev = self.device
        return ops.weight_mask(pad_vals.to(dev, torch.int64), vis_weights.to(dev, torch.float64),
                               lang_weights.to(dev, torch.float64), g.ds_grid * g.ds_grid, max_len, round_bf16=True)

    def _vision_tensor(self, x_video):
        if isinstance(x_video, (list, tuple)):          # reference passes [(tensor, "video"), ...]
            x_video = torch.stack([v[0] if isinstance(v, (list, tuple)) else v for v in x_video])
        return x_video.to(self.device, torch.float32).contiguous()

    def forward(self, x_video, x_lang, weight_mask, attention_mask=None, y=None, keep_scale=None, layout=None, ids_host=None):
        """reference :229-256 -> (regression_output fp32 [B,V], l2_reg).  attention_mask is re-derived
        on the device from the ids (ids != 0), exactly what the reference passes in (:271).
        ``layout``: packed RowLayout from ``backbone.row_layout`` (rows without padded tails)."""
        vis = self._vision_tensor(x_video)
        ids = x_lang.to(self.device, torch.int64).contiguous()
        B = ids.shape[0]
        if self.lora is not None:        # eval mode keeps the adapters (peft eval: dropout off), like the reference's validation
            hidden, key_mask = self.lora.forward(self.backbone, vis, ids, layout, train=self.training)
        elif self.full is not None and self.training:      # full fine-tune: forward that keeps what backward needs
            hidden, key_mask = self.full.forward(vis, ids, layout, ids_host=ids_host)
        else:
            hidden, key_mask = self.backbone.forward(vis, ids, layout=layout)
        if y is None:
            y = torch.zeros(B, self.config.num_target, dtype=torch.float32, device=self.device)
        pred, terms = self.head.forward(hidden, weight_mask, y, keep_scale, layout)
        self._loss_terms = terms
        return pred, terms[1]

    def _common_step(self, batch, train: bool):
        cfg, g = self.config, self.geometry
        dev = self.device
        # unpadded (packed) rows, like the reference's flash-attn path; needs the ids on the host (no sync)
        layout = self.backbone.row_layout(batch["language"], batch["padvals"]) if self.pack_tokens else None
        x_lang = batch["language"].to(dev).long()
        wm = self.make_weight_mask(batch["padvals"], batch["vis_weights"], batch["lang_weights"], x_lang.shape[1],
                                   self.nnmodule.config.tokenizer_model_max_length)
        y = batch["timeseries"].to(dev, torch.float32).to(torch.bfloat16).float().contiguous()   # reference :288
        keep = None
        if train and cfg.dropout_rate > 0:          # nn.Dropout(p) in training mode (reference :226,251)
            keep = ops.dropout_keep_scale(x_lang.shape[0], g.dim, cfg.dropout_rate, self._dropout_seed(), dev)
        ids_host = batch["language"] if batch["language"].device.type == "cpu" else None
        pred, _ = self.forward(batch["vision"], x_lang, wm, y=y, keep_scale=keep, layout=layout, ids_host=ids_host)
        return pred, y, self._loss_terms

    def prefetch_vision(self, batch, ready_event=None):
        """Start the frozen vision side (CLIP tower + STC connector) of a FUTURE batch on a side stream, so that it runs under
        the current step's decoder work; the step that later receives this batch picks the result up (Backbone.video_tokens).
        Full fine-tune: the connector trains, so only the CLIP tower runs ahead.  Called by DevicePrefetcher for batch i+1
        before step i is issued."""
        if "vision" not in batch or not batch["vision"].is_cuda:
            return
        # Deferred: the next training_step launches it right behind its forward pass, so the side stream runs under the BACKWARD
        # pass (the skinny LoRA kernels and the GEMM tails leave CUs idle there; the forward's GEMMs do not, and the bench's
        # dominant-kernel timing stays undisturbed).  A batch whose own step comes first is computed in line by that step.
        self.backbone.defer_video_tokens(batch["vision"], ready_event, tower_only=self.full is not None)

    def discard_prefetched_vision(self, batch=None):
        """A batch announced through prefetch_vision will not be consumed (``None``: none of the announced ones will)."""
        vis = None if batch is None else batch.get("vision")
        if batch is None or torch.is_tensor(vis):
            self.backbone.discard_video_tokens(vis)

    def training_step(self, batch):
        """reference :259-306.  Leaves gradients in ``.grad`` of the trainable masters."""
        self.train(True)
        self._step += 1
        if self.lora is not None:
            self.lora.rank = self.rank
        pred, y, terms = self._common_step(batch, train=True)
        self.backbone.launch_deferred_video_tokens()         # a future batch's frozen vision side: behind this forward, under this backward
        need_dh = self.lora is not None or self.full is not None
        inv_world = 1.0 / self.world_size
        dh = self.head.backward(need_dhidden=need_dh, loss_scale=inv_world, l2_scale=inv_world)
        if self.lora is not None:
            self.lora.backward(self.backbone, dh)
        elif self.full is not None:
            self.full.backward(dh)
        self._attach_gradients()
        self.log("train/brain_loss", terms[2])
        if _LP is None:
            return terms[2]          # no Lightning in this process: the plain loss value, as the built-in runner reads it
        # a scalar Lightning's automatic optimisation can call .backward() on (see _ExplicitLoss)
        return _ExplicitLoss.apply(terms[2], self, *self.parameters())

    def validation_step(self, batch):
        """reference :309-342."""
        was = self.training
        self.train(False)
        pred, y, terms = self._common_step(batch, train=False)
        self.backbone.launch_deferred_video_tokens()
        self.train(was)
        loss = terms[2].clone()
        self.log("val/brain_loss", loss)
        return {"loss": loss, "brain_preds": pred.clone(), "brain_vals": y}

    # ------------------------------------------------------------------ hooks Lightning's fit loop calls (reference train.py:41-56)
    def configure_gradient_clipping(self, optimizer, gradient_clip_val=None, gradient_clip_algorithm=None):
        """``Trainer(gradient_clip_val=1)`` (config/experiment/*.yaml:46): the global-norm clip is fused into the AdamW kernel
        (device-side norm, no host sync), so the hook only hands the threshold to the optimiser."""
        if gradient_clip_algorithm not in (None, "norm"):
            raise ValueError(f"gradient_clip_algorithm={gradient_clip_algorithm!r}: the fused optimiser clips by global norm")
        opt = getattr(optimizer, "optimizer", optimizer)            # LightningOptimizer wraps the real one
        opt.max_norm = float(gradient_clip_val or 0.0)

    def transfer_batch_to_device(self, batch, device, dataloader_idx=0):
        """Pixels / targets / weights go to the GPU; the token ids and ``padvals`` (20 KB) stay on the host so the step can size
        its unpadded row layout without a device sync (the step uploads them itself)."""
        keep = ("language", "padvals")
        return {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) and k not in keep else v) for k, v in batch.items()}

    def on_fit_start(self):
        """Called by a Lightning ``Trainer.fit`` (never by the built-in runner).  What the bridge does not support is
        refused here instead of producing wrong gradients: accumulation (every step overwrites the gradient buffers) and
        anything but ONE device - Lightning's DDP / FSDP wrappers reduce autograd gradients, and the kernels' gradients
        never flow through autograd (``_ExplicitLoss.backward`` returns None); multi-GPU runs use the built-in runner's
        clip-sharded data parallelism (``torchrun ... train.py``, parallel.attach_data_parallel)."""
        tr = self.__dict__.get("_trainer") or self.__dict__.get("trainer")
        if getattr(tr, "accumulate_grad_batches", 1) not in (None, 1):
            raise ValueError("accumulate_grad_batches > 1 is not supported: every training_step overwrites the gradient buffers "
                             "(the reference's configs do not accumulate)")
        world = getattr(tr, "world_size", 1) or 1
        strategy = type(getattr(tr, "strategy", None)).__name__
        if world > 1 or strategy not in ("NoneType", "SingleDeviceStrategy"):
            raise ValueError(f"lightning.pytorch.Trainer with world_size={world}, strategy={strategy}: the Lightning bridge drives "
                             "ONE device.  For N GPUs launch `torchrun --nproc-per-node N train.py ...` with VLB_TRAINER=builtin "
                             "(the default): gradients are reduce-scattered by the package's own data-parallel step")

    def on_save_checkpoint(self, checkpoint: dict) -> None:
        """Lightning hook: the counter-based dropout state (head step, LoRA step) next to ``state_dict`` /
        ``optimizer_states`` (VlbAdamW.state_dict carries moments, step count and the full fine-tune's backbone masters)."""
        checkpoint["vlb_rng"] = self.rng_state()

    def on_load_checkpoint(self, checkpoint: dict) -> None:
        if "vlb_rng" in checkpoint:
            self.set_rng_state(checkpoint["vlb_rng"])

    def state_dict(self, *args, **kwargs):
        """Trainables only, upstream / peft names (what a Lightning ModelCheckpoint stores for this module; the reference saves
        the whole frozen 7B and notes the TODO, train.py:60)."""
        if getattr(self, "nnmodule", None) is None:
            return {}
        return self.trainable_state_dict()

    def load_state_dict(self, state_dict, strict: bool = True, **kwargs):
        self.load_trainable_state_dict(state_dict)

    # ------------------------------------------------------------------ optimiser
    def configure_optimizers(self):
        """reference :345-379: AdamW over the trainables + CosineAnnealingLR stepped every step."""
        cfg = self.config
        from .flat import FlatTrainables
        if getattr(self, "flat", None) is None:
            self.flat = FlatTrainables(self)          # masters / bf16 copies / grads / moments -> flat buffers
        named = self.trainable_named_parameters()
        flats = [self.flat] + ([self.full.flat] if self.full is not None else [])
        self.optimizer = VlbAdamW(named, flats, lr=cfg.lr, betas=tuple(cfg.betas), eps=cfg.eps,
                                  weight_decay=cfg.weight_decay, max_norm=cfg.gradient_clip_val)
        if self.lora is not None:
            self.optimizer.post_step.append(self.lora.refresh)
        if self.full is not None:
            self.optimizer.post_step.append(self.full.refresh_transposed)
        self.lr_scheduler_args = {"last_epoch": cfg.last_epoch, "T_max": cfg.t_max}
        self.scheduler = getattr(torch.optim.lr_scheduler, cfg.lr_scheduler_name)(self.optimizer, **self.lr_scheduler_args)
        return [self.optimizer], [{"scheduler": self.scheduler, "interval": "step", "frequency": 1}]


def load_safetensors_dir(path: str) -> dict:
    """Load every *.safetensors shard of a local checkpoint directory (upstream naming)."""
    from safetensors.torch import load_file
    sd = {}
    for f in sorted(os.listdir(path)):
        if f.endswith(".safetensors"):
            sd.update(load_file(os.path.join(path, f)))
    if not sd:
        raise FileNotFoundError(f"no *.safetensors under {path}")
    return sd
