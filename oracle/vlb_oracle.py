"""CPU oracle for the phantom_vlb fine-tuning hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``phantom_vlb_amd``) never imports anything under ``oracle/`` and
fails loudly when the HIP library is missing.

It is a plain-PyTorch, fp32, CPU restatement of everything
``VLBLitModule.training_step`` computes (reference
``src/litmodule/videollama2_vlb_litmodule.py:259-306``), including the un-vendored
third-party arithmetic the reference reaches through ``self.nnmodule(...)``:

  stage                                   follows
  --------------------------------------  ------------------------------------------------
  make_weight_mask                        src/litmodule/videollama2_vlb_litmodule.py:178-203
  clip_tower                              transformers modeling_clip.py:138-219,280-384,594-657
                                          (HF CLIPVisionModel, layer -2, CLS dropped)
  stc_connector                           VideoLLaMA2 projector.STCConnector + timm RegStage
                                          (source ABSENT from /root/reference: restated from the
                                          published architecture, SURVEY.md Appendix B - UNPINNED)
  splice_multimodal                       VideoLLaMA2 prepare_inputs_labels_for_multimodal
                                          (ABSENT, SURVEY.md Appendix B - UNPINNED)
  mistral_decoder (+LoRA)                 transformers modeling_mistral.py:35-240,262-316,386;
                                          peft LoRA y += (alpha/r) B(A(drop(x))) (peft ABSENT - UNPINNED)
  brain_head                              src/litmodule/...:229-256 ; src/utils.py:40-73
  training_loss                           src/litmodule/...:288-302
  adamw_cosine_step                       src/litmodule/...:345-379 + Trainer gradient_clip_val

Pin status (see oracle/gen_golden.py, which produced tests/golden/*):
  * brain_head pooling / ridge: PINNED against the reference's own
    ``src/utils.py`` HRFConvolveLayer / RidgeRegressionLayer imported from
    /root/reference in the build container.
  * clip_tower, mistral_decoder: PINNED against the installed ``transformers``
    (5.15.0; reference pins 4.53.3) CLIPVisionModel / MistralModel on random-init
    mini configs.
  * stc_connector, splice, LoRA, make_weight_mask layout: **parity unpinned** -
    the reference ships no tests and their sources are not
    in the snapshot; they rest on this restatement plus hand-checked vectors.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F

VIDEO_TOKEN_ID = -201  # src/preprocessing/videollama2_vlb_extractfeatures.py:235-236


# --------------------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------------------
@dataclass
class Geometry:
    """Every shape constant the path hard-codes, parameterised."""
    # clips
    num_frames: int = 12
    image_size: int = 336
    patch: int = 14
    # CLIP tower
    vit_dim: int = 1024
    vit_layers: int = 24
    vit_heads: int = 16
    vit_ff: int = 4096
    vit_eps: float = 1e-5
    vit_select_layer: int = -2
    # STC connector
    proj_depth: int = 4
    proj_eps: float = 1e-6      # timm LayerNorm2d default
    proj_se_ratio: float = 0.25
    # decoder
    dim: int = 4096
    layers: int = 32
    heads: int = 32
    kv_heads: int = 8
    head_dim: int = 128
    ff: int = 14336
    vocab: int = 32000
    rms_eps: float = 1e-5
    rope_theta: float = 1e6
    max_len: int = 2048         # tokenizer_model_max_length
    # head
    num_target: int = 2048
    ln_eps: float = 1e-5
    l2_lambda: float = 1e-3
    # lora
    lora_r: int = 16
    lora_alpha: int = 32

    @property
    def grid(self) -> int:
        return self.image_size // self.patch

    @property
    def ds_frames(self) -> int:      # Conv3d k=2 s=2 p=1
        return (self.num_frames + 2 - 2) // 2 + 1

    @property
    def ds_grid(self) -> int:
        return (self.grid + 2 - 2) // 2 + 1

    @property
    def vis_tokens(self) -> int:
        return self.ds_frames * self.ds_grid * self.ds_grid

    @property
    def lang_len(self) -> int:       # src/litmodule/...:180-181 identity
        return self.max_len - self.vis_tokens + 1

    @property
    def vit_layers_run(self) -> int:  # hidden_states[-2] needs layers[:-1]
        return self.vit_layers + 1 + self.vit_select_layer


def geometry_7b(**kw) -> Geometry:
    return Geometry(**kw)


def geometry_mini(**kw) -> Geometry:
    """BASELINE.json configs[0]: 2-layer mini-VideoLLaMA2, 128-voxel head, 8-frame clips.

    Head sizes are the production ones (ViT hd=64, decoder hd=128, GQA) so the mini
    config exercises the same kernel instantiations as the 7B one.
    """
    g = dict(num_frames=8, image_size=84, patch=14,
             vit_dim=128, vit_layers=3, vit_heads=2, vit_ff=256,
             dim=512, layers=2, heads=4, kv_heads=1, head_dim=128, ff=1024, vocab=512,
             max_len=128, num_target=128)
    g.update(kw)
    return Geometry(**g)


LORA_TARGETS = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")


# --------------------------------------------------------------------------------------
# parameter construction (random init with the initialisers the real classes use)
# --------------------------------------------------------------------------------------
def _regstage_shapes(prefix, cin, cout, depth, se_ratio):
    shapes = {}
    for b in range(depth):
        ci = cin if b == 0 else cout
        p = f"{prefix}.b{b + 1}"
        rd = int(round(ci * se_ratio))
        shapes[f"{p}.conv1.conv.weight"] = (cout, ci, 1, 1)
        shapes[f"{p}.conv1.bn.weight"] = (cout,)
        shapes[f"{p}.conv1.bn.bias"] = (cout,)
        shapes[f"{p}.conv2.conv.weight"] = (cout, 1, 3, 3)
        shapes[f"{p}.conv2.bn.weight"] = (cout,)
        shapes[f"{p}.conv2.bn.bias"] = (cout,)
        shapes[f"{p}.se.fc1.weight"] = (rd, cout, 1, 1)
        shapes[f"{p}.se.fc1.bias"] = (rd,)
        shapes[f"{p}.se.fc2.weight"] = (cout, rd, 1, 1)
        shapes[f"{p}.se.fc2.bias"] = (cout,)
        shapes[f"{p}.conv3.conv.weight"] = (cout, cout, 1, 1)
        shapes[f"{p}.conv3.bn.weight"] = (cout,)
        shapes[f"{p}.conv3.bn.bias"] = (cout,)
        if ci != cout:
            shapes[f"{p}.downsample.conv.weight"] = (cout, ci, 1, 1)
            shapes[f"{p}.downsample.bn.weight"] = (cout,)
            shapes[f"{p}.downsample.bn.bias"] = (cout,)
    return shapes


def param_shapes(g: Geometry, lora: bool = False) -> dict[str, tuple]:
    """Name -> shape for every tensor of the model, upstream state-dict naming."""
    s: dict[str, tuple] = {}
    v = "model.vision_tower.vision_tower.vision_model"
    s[f"{v}.embeddings.class_embedding"] = (g.vit_dim,)
    s[f"{v}.embeddings.patch_embedding.weight"] = (g.vit_dim, 3, g.patch, g.patch)
    s[f"{v}.embeddings.position_embedding.weight"] = (g.grid * g.grid + 1, g.vit_dim)
    s[f"{v}.pre_layrnorm.weight"] = (g.vit_dim,)
    s[f"{v}.pre_layrnorm.bias"] = (g.vit_dim,)
    for i in range(g.vit_layers):
        p = f"{v}.encoder.layers.{i}"
        for ln in ("layer_norm1", "layer_norm2"):
            s[f"{p}.{ln}.weight"] = (g.vit_dim,)
            s[f"{p}.{ln}.bias"] = (g.vit_dim,)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[f"{p}.self_attn.{n}.weight"] = (g.vit_dim, g.vit_dim)
            s[f"{p}.self_attn.{n}.bias"] = (g.vit_dim,)
        s[f"{p}.mlp.fc1.weight"] = (g.vit_ff, g.vit_dim)
        s[f"{p}.mlp.fc1.bias"] = (g.vit_ff,)
        s[f"{p}.mlp.fc2.weight"] = (g.vit_dim, g.vit_ff)
        s[f"{p}.mlp.fc2.bias"] = (g.vit_dim,)
    m = "model.mm_projector"
    s.update(_regstage_shapes(f"{m}.s1", g.vit_dim, g.dim, g.proj_depth, g.proj_se_ratio))
    s[f"{m}.sampler.0.weight"] = (g.dim, g.dim, 2, 2, 2)
    s[f"{m}.sampler.0.bias"] = (g.dim,)
    s.update(_regstage_shapes(f"{m}.s2", g.dim, g.dim, g.proj_depth, g.proj_se_ratio))
    s[f"{m}.readout.0.weight"] = (g.dim, g.dim)
    s[f"{m}.readout.0.bias"] = (g.dim,)
    s[f"{m}.readout.2.weight"] = (g.dim, g.dim)
    s[f"{m}.readout.2.bias"] = (g.dim,)
    s["model.embed_tokens.weight"] = (g.vocab, g.dim)
    for i in range(g.layers):
        p = f"model.layers.{i}"
        lin = {
            "self_attn.q_proj": (g.heads * g.head_dim, g.dim),
            "self_attn.k_proj": (g.kv_heads * g.head_dim, g.dim),
            "self_attn.v_proj": (g.kv_heads * g.head_dim, g.dim),
            "self_attn.o_proj": (g.dim, g.heads * g.head_dim),
            "mlp.gate_proj": (g.ff, g.dim),
            "mlp.up_proj": (g.ff, g.dim),
            "mlp.down_proj": (g.dim, g.ff),
        }
        for n, shp in lin.items():
            s[f"{p}.{n}.weight"] = shp
            if lora:
                s[f"{p}.{n}.lora_A.weight"] = (g.lora_r, shp[1])
                s[f"{p}.{n}.lora_B.weight"] = (shp[0], g.lora_r)
        s[f"{p}.input_layernorm.weight"] = (g.dim,)
        s[f"{p}.post_attention_layernorm.weight"] = (g.dim,)
    s["model.norm.weight"] = (g.dim,)
    # brain head, attribute names of src/litmodule/...:210-226
    s["layer_norm1.weight"] = (g.dim,)
    s["layer_norm1.bias"] = (g.dim,)
    s["layer_norm2.weight"] = (g.dim,)
    s["layer_norm2.bias"] = (g.dim,)
    s["ridge_layer.linear.weight"] = (g.num_target, g.dim)
    s["ridge_layer.linear.bias"] = (g.num_target,)
    return s


def init_params(g: Geometry, seed: int = 1234, lora: bool = False,
                lora_b_std: float = 0.0, dtype=torch.float32) -> dict[str, torch.Tensor]:
    """Random-init weights with HF/timm/peft-style initialisers (std 0.02 linears,
    ones/zeros norms, LoRA A kaiming-uniform(a=sqrt 5) and B zeros unless lora_b_std>0,
    nn.Linear default init for the ridge head)."""
    gen = torch.Generator().manual_seed(seed)
    out = {}
    for name, shp in param_shapes(g, lora).items():
        if name.endswith("lora_A.weight"):
            bound = 1.0 / math.sqrt(shp[1])  # kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), ..)
            t = (torch.rand(shp, generator=gen) * 2 - 1) * bound
        elif name.endswith("lora_B.weight"):
            t = torch.randn(shp, generator=gen) * lora_b_std if lora_b_std > 0 else torch.zeros(shp)
        elif name.startswith("ridge_layer.linear"):
            bound = 1.0 / math.sqrt(g.dim)
            t = (torch.rand(shp, generator=gen) * 2 - 1) * bound
        elif name.startswith("layer_norm") or "layernorm" in name or "layer_norm" in name \
                or name.endswith("norm.weight") or ".bn." in name or "pre_layrnorm" in name:
            if name.endswith("bias"):
                t = torch.randn(shp, generator=gen) * 0.02   # non-trivial affine so parity tests see it
            else:
                t = 1.0 + torch.randn(shp, generator=gen) * 0.02
        elif name.endswith(".bias"):
            t = torch.randn(shp, generator=gen) * 0.02
        elif "conv2.conv.weight" in name:
            t = torch.randn(shp, generator=gen) * (1.0 / 3.0)   # depthwise 3x3, fan_in 9
        elif "mm_projector" in name and name.endswith("weight"):
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            t = torch.randn(shp, generator=gen) * (1.0 / math.sqrt(fan_in))
        elif "class_embedding" in name:
            t = torch.randn(shp, generator=gen) * 0.02
        else:
            t = torch.randn(shp, generator=gen) * 0.02
        out[name] = t.to(dtype)
    return out


# --------------------------------------------------------------------------------------
# synthetic batch
# --------------------------------------------------------------------------------------
GLOVER_VIS_WEIGHTS_7 = (0.0463, 0.0644, 0.0829, 0.0964, 0.0982, 0.0826, 0.0491)


def synthetic_batch(g: Geometry, batch: int, seed: int = 1234, inst_len: int = 9) -> dict[str, torch.Tensor]:
    """One batch in the lazy-load sample schema (src/datamodule/...:98-109 after collate)."""
    gen = torch.Generator().manual_seed(seed)
    L = g.lang_len
    vision = torch.randn(batch, g.num_frames, 3, g.image_size, g.image_size, generator=gen)
    language = torch.zeros(batch, L)
    padvals = torch.zeros(batch, 3, dtype=torch.int64)
    n_vis = g.ds_frames
    base = torch.tensor(GLOVER_VIS_WEIGHTS_7, dtype=torch.float64)
    vis_w = torch.stack([base[torch.arange(n_vis) % 7] * (1 + 0.05 * b) for b in range(batch)])
    lang_w = torch.zeros(batch, 64, dtype=torch.float64)
    max_dialog = min(58, L - (2 + inst_len + 4) - 2)
    for b in range(batch):
        dialog_len = int(torch.randint(0, max_dialog + 1, (1,), generator=gen))
        body = 2 + inst_len + dialog_len + 4
        max_pad = L - 1 - body - 1
        pad_len = int(torch.randint(0, min(300, max_pad) + 1, (1,), generator=gen))
        P = L - 1 - body - pad_len                      # prompt tokens before <video>
        ids = torch.randint(3, g.vocab, (L,), generator=gen).float()
        ids[P] = VIDEO_TOKEN_ID
        if pad_len:
            ids[L - pad_len:] = 0
        language[b] = ids
        padvals[b] = torch.tensor([pad_len, inst_len, dialog_len])
        lang_w[b, :dialog_len] = torch.rand(dialog_len, generator=gen, dtype=torch.float64) * 0.2
    timeseries = torch.randn(batch, g.num_target, generator=gen)
    return dict(vision=vision, language=language, timeseries=timeseries,
                padvals=padvals, vis_weights=vis_w, lang_weights=lang_w)


# --------------------------------------------------------------------------------------
# a2: weight mask   (src/litmodule/videollama2_vlb_litmodule.py:178-203)
# --------------------------------------------------------------------------------------
def make_weight_mask(pad_vals, vis_weights, lang_weights, lang_len, max_len, tokens_per_frame=13 * 13):
    """Row b = [left zeros][each vis weight x tokens_per_frame][2+inst zeros][dialog lang weights][4+pad zeros]."""
    n_vis = vis_weights.shape[1]
    feature_len = n_vis * tokens_per_frame + lang_len - 1
    assert feature_len == max_len
    rows = []
    for i in range(pad_vals.shape[0]):
        pad_len, inst_len, dialog_len = (int(x) for x in pad_vals[i])
        trial = torch.cat([
            vis_weights[i].repeat_interleave(tokens_per_frame).float(),
            torch.zeros(2 + inst_len),
            lang_weights[i][:dialog_len].float(),
            torch.zeros(4 + pad_len),
        ])
        rows.append(torch.cat([torch.zeros(feature_len - trial.shape[0]), trial]))
    return torch.stack(rows)


# --------------------------------------------------------------------------------------
# a4: CLIP tower
# --------------------------------------------------------------------------------------
def _quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def clip_tower(p, pixels, g: Geometry, prefix="model.vision_tower.vision_tower.vision_model"):
    """pixels (N,3,H,W) -> patch tokens of hidden_states[select_layer] (N, grid^2, vit_dim)."""
    N = pixels.shape[0]
    x = F.conv2d(pixels, p[f"{prefix}.embeddings.patch_embedding.weight"], stride=g.patch)
    x = x.flatten(2).transpose(1, 2)
    cls = p[f"{prefix}.embeddings.class_embedding"].expand(N, 1, -1)
    x = torch.cat([cls, x], 1) + p[f"{prefix}.embeddings.position_embedding.weight"][None]
    x = F.layer_norm(x, (g.vit_dim,), p[f"{prefix}.pre_layrnorm.weight"], p[f"{prefix}.pre_layrnorm.bias"], g.vit_eps)
    hd = g.vit_dim // g.vit_heads
    for i in range(g.vit_layers_run):
        lp = f"{prefix}.encoder.layers.{i}"
        h = F.layer_norm(x, (g.vit_dim,), p[f"{lp}.layer_norm1.weight"], p[f"{lp}.layer_norm1.bias"], g.vit_eps)
        q = F.linear(h, p[f"{lp}.self_attn.q_proj.weight"], p[f"{lp}.self_attn.q_proj.bias"])
        k = F.linear(h, p[f"{lp}.self_attn.k_proj.weight"], p[f"{lp}.self_attn.k_proj.bias"])
        v = F.linear(h, p[f"{lp}.self_attn.v_proj.weight"], p[f"{lp}.self_attn.v_proj.bias"])
        sh = (N, -1, g.vit_heads, hd)
        q, k, v = (t.view(sh).transpose(1, 2) for t in (q, k, v))
        a = torch.softmax(q @ k.transpose(2, 3) * hd ** -0.5, -1) @ v
        a = a.transpose(1, 2).reshape(N, -1, g.vit_dim)
        x = x + F.linear(a, p[f"{lp}.self_attn.out_proj.weight"], p[f"{lp}.self_attn.out_proj.bias"])
        h = F.layer_norm(x, (g.vit_dim,), p[f"{lp}.layer_norm2.weight"], p[f"{lp}.layer_norm2.bias"], g.vit_eps)
        h = _quick_gelu(F.linear(h, p[f"{lp}.mlp.fc1.weight"], p[f"{lp}.mlp.fc1.bias"]))
        x = x + F.linear(h, p[f"{lp}.mlp.fc2.weight"], p[f"{lp}.mlp.fc2.bias"])
    return x[:, 1:]


# --------------------------------------------------------------------------------------
# a5: STC connector (timm RegStage x2 around Conv3d, then MLP readout)
# --------------------------------------------------------------------------------------
def _ln2d(x, w, b, eps):
    return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), w, b, eps).permute(0, 3, 1, 2)


def _regnet_bottleneck(p, pre, x, g: Geometry):
    """timm.models.regnet.Bottleneck(bottle_ratio=1, group_size=1 -> depthwise, se_ratio=.25,
    act=SiLU, norm=LayerNorm2d, downsample='conv1x1' on channel change, identity otherwise)."""
    sc = x
    c = p[f"{pre}.conv1.conv.weight"].shape[0]
    x = F.silu(_ln2d(F.conv2d(x, p[f"{pre}.conv1.conv.weight"]), p[f"{pre}.conv1.bn.weight"], p[f"{pre}.conv1.bn.bias"], g.proj_eps))
    x = F.silu(_ln2d(F.conv2d(x, p[f"{pre}.conv2.conv.weight"], padding=1, groups=c), p[f"{pre}.conv2.bn.weight"], p[f"{pre}.conv2.bn.bias"], g.proj_eps))
    se = x.mean((2, 3), keepdim=True)
    se = F.silu(F.conv2d(se, p[f"{pre}.se.fc1.weight"], p[f"{pre}.se.fc1.bias"]))
    se = F.conv2d(se, p[f"{pre}.se.fc2.weight"], p[f"{pre}.se.fc2.bias"])
    x = x * torch.sigmoid(se)
    x = _ln2d(F.conv2d(x, p[f"{pre}.conv3.conv.weight"]), p[f"{pre}.conv3.bn.weight"], p[f"{pre}.conv3.bn.bias"], g.proj_eps)
    if f"{pre}.downsample.conv.weight" in p:
        sc = _ln2d(F.conv2d(sc, p[f"{pre}.downsample.conv.weight"]), p[f"{pre}.downsample.bn.weight"], p[f"{pre}.downsample.bn.bias"], g.proj_eps)
    return F.silu(x + sc)


def stc_connector(p, feats, g: Geometry, prefix="model.mm_projector"):
    """feats (B,T,grid^2,vit_dim) -> (B, ds_frames*ds_grid^2, dim)."""
    B, T = feats.shape[:2]
    hw = g.grid
    x = feats.view(B * T, hw, hw, g.vit_dim).permute(0, 3, 1, 2)              # (b t) d h w
    for b in range(g.proj_depth):
        x = _regnet_bottleneck(p, f"{prefix}.s1.b{b + 1}", x, g)
    x = x.view(B, T, g.dim, hw, hw).permute(0, 2, 1, 3, 4)                   # b d t h w
    x = F.silu(F.conv3d(x, p[f"{prefix}.sampler.0.weight"], p[f"{prefix}.sampler.0.bias"], stride=2, padding=1))
    nt, nh = x.shape[2], x.shape[3]
    x = x.permute(0, 2, 1, 3, 4).reshape(B * nt, g.dim, nh, nh)
    for b in range(g.proj_depth):
        x = _regnet_bottleneck(p, f"{prefix}.s2.b{b + 1}", x, g)
    x = x.view(B, nt, g.dim, nh * nh).permute(0, 1, 3, 2).reshape(B, nt * nh * nh, g.dim)   # b (t h w) d
    x = F.linear(x, p[f"{prefix}.readout.0.weight"], p[f"{prefix}.readout.0.bias"])
    x = F.gelu(x)
    return F.linear(x, p[f"{prefix}.readout.2.weight"], p[f"{prefix}.readout.2.bias"])


# --------------------------------------------------------------------------------------
# a6: token splice
# --------------------------------------------------------------------------------------
def splice_multimodal(embed_w, ids, video_tokens):
    """ids (B,L) with exactly one VIDEO_TOKEN_ID per row, video_tokens (B,Nv,D)
    -> inputs_embeds (B, L-1+Nv, D), key mask (B, L-1+Nv) bool.
    Padding ids (0) are embedded, not stripped; the mask is ids!=0 left-extended with ones."""
    B, L = ids.shape
    Nv = video_tokens.shape[1]
    rows, masks = [], []
    for b in range(B):
        pos = int((ids[b] == VIDEO_TOKEN_ID).nonzero()[0])
        left = F.embedding(ids[b, :pos], embed_w)
        right = F.embedding(ids[b, pos + 1:], embed_w)
        rows.append(torch.cat([left, video_tokens[b], right], 0))
        m = ids[b] != 0
        masks.append(torch.cat([torch.ones(Nv - 1, dtype=torch.bool), m]))
    return torch.stack(rows), torch.stack(masks)


# --------------------------------------------------------------------------------------
# a7/a8/a14: Mistral decoder with optional LoRA
# --------------------------------------------------------------------------------------
def _rms(x, w, eps):
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def rope_tables(g: Geometry, S: int):
    inv = 1.0 / (g.rope_theta ** (torch.arange(0, g.head_dim, 2, dtype=torch.float32) / g.head_dim))
    fr = torch.arange(S, dtype=torch.float32)[:, None] * inv[None]
    emb = torch.cat([fr, fr], -1)
    return emb.cos(), emb.sin()


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], -1)


def _lin(p, name, x, g: Geometry, lora_drop=None):
    y = F.linear(x, p[f"{name}.weight"])
    a = p.get(f"{name}.lora_A.weight")
    if a is not None:
        xd = x if lora_drop is None else x * lora_drop[name]
        y = y + (g.lora_alpha / g.lora_r) * F.linear(F.linear(xd, a), p[f"{name}.lora_B.weight"])
    return y


def mistral_decoder(p, x, key_mask, g: Geometry, return_layers=False, lora_drop=None):
    """x (B,S,D) inputs_embeds, key_mask (B,S) bool -> post-final-norm hidden (B,S,D)
    (== HF hidden_states[-1], modeling_mistral.py:386)."""
    B, S, _ = x.shape
    cos, sin = rope_tables(g, S)
    causal = torch.ones(S, S, dtype=torch.bool).tril()
    allow = causal[None, None] & key_mask[:, None, None, :]
    bias = torch.zeros(B, 1, S, S).masked_fill(~allow, torch.finfo(torch.float32).min)
    rep = g.heads // g.kv_heads
    outs = []
    for i in range(g.layers):
        lp = f"model.layers.{i}"
        h = _rms(x, p[f"{lp}.input_layernorm.weight"], g.rms_eps)
        q = _lin(p, f"{lp}.self_attn.q_proj", h, g, lora_drop).view(B, S, g.heads, g.head_dim).transpose(1, 2)
        k = _lin(p, f"{lp}.self_attn.k_proj", h, g, lora_drop).view(B, S, g.kv_heads, g.head_dim).transpose(1, 2)
        v = _lin(p, f"{lp}.self_attn.v_proj", h, g, lora_drop).view(B, S, g.kv_heads, g.head_dim).transpose(1, 2)
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        k = k.repeat_interleave(rep, 1)
        v = v.repeat_interleave(rep, 1)
        a = torch.softmax(q @ k.transpose(2, 3) * g.head_dim ** -0.5 + bias, -1) @ v
        a = a.transpose(1, 2).reshape(B, S, g.heads * g.head_dim)
        x = x + _lin(p, f"{lp}.self_attn.o_proj", a, g, lora_drop)
        h = _rms(x, p[f"{lp}.post_attention_layernorm.weight"], g.rms_eps)
        gate = _lin(p, f"{lp}.mlp.gate_proj", h, g, lora_drop)
        up = _lin(p, f"{lp}.mlp.up_proj", h, g, lora_drop)
        x = x + _lin(p, f"{lp}.mlp.down_proj", F.silu(gate) * up, g, lora_drop)
        outs.append(x)
    x = _rms(x, p["model.norm.weight"], g.rms_eps)
    return (x, outs) if return_layers else x


# --------------------------------------------------------------------------------------
# a9-a13: brain head + loss
# --------------------------------------------------------------------------------------
def hrf_convolve(embeddings, hrf_weights):
    """HRFConvolveLayer.forward (src/utils.py:44-56): HRF-weighted sum of the token embeddings."""
    return torch.einsum("bse,bs->be", embeddings, hrf_weights)


def ridge_regression(weight, bias, x, l2_lambda, add_regularization=True):
    """RidgeRegressionLayer.forward (src/utils.py:59-73): Linear(x) and l2_lambda * ||W||_F^2 (bias not penalised)."""
    out = F.linear(x, weight, bias)
    return (out, l2_lambda * weight.pow(2).sum()) if add_regularization else out


def brain_head(p, hidden, weight_mask, g: Geometry, keep_mask=None, dropout_p=0.0):
    """LN1 -> einsum('bse,bs->be') -> LN2 -> dropout -> Linear ; l2 = lambda*||W||_F^2."""
    h = F.layer_norm(hidden, (g.dim,), p["layer_norm1.weight"], p["layer_norm1.bias"], g.ln_eps)
    pooled = hrf_convolve(h, weight_mask)
    z = F.layer_norm(pooled, (g.dim,), p["layer_norm2.weight"], p["layer_norm2.bias"], g.ln_eps)
    if keep_mask is not None:
        z = z * keep_mask / (1.0 - dropout_p)
    pred, l2 = ridge_regression(p["ridge_layer.linear.weight"], p["ridge_layer.linear.bias"], z, g.l2_lambda)
    return pred, l2, dict(ln1=h, pooled=pooled, ln2=z)


def round_bf16(p: dict) -> dict:
    """The reference holds every parameter in bf16 (torch_dtype=bf16, head dtype=bf16;
    src/litmodule/...:155,211-225): the oracle keeps fp32 MATH on those bf16-valued weights."""
    return {k: v.detach().to(torch.bfloat16).float() for k, v in p.items()}


def backbone_forward(p, batch, g: Geometry, stages=None, lora_drop=None):
    """vision/language of a batch -> post-norm hidden (B,max_len,dim) and key mask.
    Pixels are rounded to bf16 first, as training_step does (src/litmodule/...:267)."""
    B = batch["vision"].shape[0]
    pix = batch["vision"].to(torch.bfloat16).float().reshape(B * g.num_frames, 3, g.image_size, g.image_size)
    vit = clip_tower(p, pix, g).view(B, g.num_frames, g.grid * g.grid, g.vit_dim)
    vid = stc_connector(p, vit, g)
    ids = batch["language"].long()
    emb, key_mask = splice_multimodal(p["model.embed_tokens.weight"], ids, vid)
    hidden, layers = mistral_decoder(p, emb, key_mask, g, return_layers=True, lora_drop=lora_drop)
    if stages is not None:
        stages.update(vit_tokens=vit, video_tokens=vid, inputs_embeds=emb, key_mask=key_mask,
                      layer_outputs=layers, hidden=hidden)
    return hidden, key_mask


def training_loss(p, batch, g: Geometry, keep_mask=None, dropout_p=0.0, stages=None, lora_drop=None):
    """Full restatement of training_step -> (brain_loss, pred)."""
    hidden, _ = backbone_forward(p, batch, g, stages, lora_drop)
    wm = make_weight_mask(batch["padvals"], batch["vis_weights"], batch["lang_weights"],
                          batch["language"].shape[1], g.max_len, g.ds_grid * g.ds_grid)
    wm = wm.to(torch.bfloat16).float()                       # the reference builds the mask in bf16 (:190-194)
    y = batch["timeseries"].to(torch.bfloat16).float()       # :288
    pred, l2, hs = brain_head(p, hidden, wm, g, keep_mask, dropout_p)
    loss = F.mse_loss(pred, y) + l2
    if stages is not None:
        stages.update(weight_mask=wm, pred=pred, l2=l2, loss=loss, **{f"head_{k}": v for k, v in hs.items()})
    return loss, pred


def trainable_names(p, freeze_backbone: bool, use_lora: bool):
    """src/litmodule/...:86-120: head always trains; LoRA A/B when use_lora; nothing else in configs 2-4."""
    head = [n for n in p if n.startswith(("layer_norm1", "layer_norm2", "ridge_layer"))]
    if use_lora:
        head += [n for n in p if ".lora_" in n]
    return head


# --------------------------------------------------------------------------------------
# a16: optimiser step (AdamW + cosine, global-norm clip 1.0 as Lightning applies it)
# --------------------------------------------------------------------------------------
def clip_grad_norm(grads: dict, max_norm: float = 1.0):
    total = torch.sqrt(sum(gr.double().pow(2).sum() for gr in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return {k: v * coef for k, v in grads.items()}, total


def cosine_lr(base_lr: float, step: int, t_max: int, eta_min: float = 0.0):
    """Closed form of torch CosineAnnealingLR(T_max) after `step` scheduler steps."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * step / t_max)) / 2


def adamw_step(param, grad, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, wd=1e-2):
    """torch.optim.AdamW single-tensor update, fp32; step counts from 1."""
    param = param * (1 - lr * wd)
    m = beta1 * m + (1 - beta1) * grad
    v = beta2 * v + (1 - beta2) * grad * grad
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    param = param - (lr / bc1) * m / denom
    return param, m, v
