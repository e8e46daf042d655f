"""VideoLLaMA2 backbone (CLIP tower -> STC connector -> token splice -> Mistral decoder) on libvlb.

Host-side mirror of what the reference reaches through ``self.nnmodule(input_ids, attention_mask,
output_hidden_states=True, images=x_video)`` (src/litmodule/videollama2_vlb_litmodule.py:231-236):
the un-vendored ``Videollama2MistralForCausalLM``.  Only the piece the loss reads is computed -
``hidden_states[-1]`` (post final RMSNorm); the lm_head logits and the 33 hidden-state copies the
reference materialises are never used by the loss and are dropped.

Layout (MI355X-first): every activation is a 2-D [tokens, channels] bf16 matrix (channels-last),
so 1x1 convs, the patch conv, Conv3d (via im2col) and all linears are one TN GEMM kernel;
q/k/v and gate/up are fused into single weights; frozen weights also keep a transposed copy for
the dgrad GEMMs of the LoRA configuration (288 GB of HBM makes that free).
Weights are addressed by their upstream state-dict names so real checkpoints can be loaded.
"""
from __future__ import annotations

import math

import torch

from . import ops
from .geometry import Geometry, VIDEO_TOKEN_ID

BF16 = torch.bfloat16
V_PRE = "model.vision_tower.vision_tower.vision_model"
M_PRE = "model.mm_projector"


def _bf(t, dev):
    return t.detach().to(device=dev, dtype=BF16).contiguous()


class Weights:
    """Kernel-ready bf16 weights built from an upstream-named state dict (any dtype / device)."""

    def __init__(self, g: Geometry, sd: dict, device, keep_transposed: bool = False, gate_up_interleaved: bool | None = None):
        self.g = g
        self.dev = device
        d = device
        # ---- CLIP tower
        w = sd[f"{V_PRE}.embeddings.patch_embedding.weight"].reshape(g.vit_dim, -1)
        wp = torch.zeros(g.vit_dim, g.patch_k_padded, dtype=w.dtype, device=w.device)
        wp[:, :g.patch_k] = w
        self.patch_w = _bf(wp, d)
        self.cls = _bf(sd[f"{V_PRE}.embeddings.class_embedding"], d)
        self.pos = _bf(sd[f"{V_PRE}.embeddings.position_embedding.weight"], d)
        self.pre_ln = (_bf(sd[f"{V_PRE}.pre_layrnorm.weight"], d), _bf(sd[f"{V_PRE}.pre_layrnorm.bias"], d))
        self.vit = []
        for i in range(g.vit_layers_run):
            p = f"{V_PRE}.encoder.layers.{i}"
            a = f"{p}.self_attn"
            self.vit.append(dict(
                ln1=(_bf(sd[f"{p}.layer_norm1.weight"], d), _bf(sd[f"{p}.layer_norm1.bias"], d)),
                ln2=(_bf(sd[f"{p}.layer_norm2.weight"], d), _bf(sd[f"{p}.layer_norm2.bias"], d)),
                wqkv=_bf(torch.cat([sd[f"{a}.q_proj.weight"], sd[f"{a}.k_proj.weight"], sd[f"{a}.v_proj.weight"]], 0), d),
                bqkv=_bf(torch.cat([sd[f"{a}.q_proj.bias"], sd[f"{a}.k_proj.bias"], sd[f"{a}.v_proj.bias"]], 0), d),
                wo=_bf(sd[f"{a}.out_proj.weight"], d), bo=_bf(sd[f"{a}.out_proj.bias"], d),
                w1=_bf(sd[f"{p}.mlp.fc1.weight"], d), b1=_bf(sd[f"{p}.mlp.fc1.bias"], d),
                w2=_bf(sd[f"{p}.mlp.fc2.weight"], d), b2=_bf(sd[f"{p}.mlp.fc2.bias"], d)))
        # ---- STC connector
        def stage(prefix):
            blocks = []
            for b in range(g.proj_depth):
                q = f"{prefix}.b{b + 1}"
                blk = dict(
                    conv1=_bf(sd[f"{q}.conv1.conv.weight"].flatten(1), d),
                    bn1=(_bf(sd[f"{q}.conv1.bn.weight"], d), _bf(sd[f"{q}.conv1.bn.bias"], d)),
                    dw=_bf(sd[f"{q}.conv2.conv.weight"].flatten(1).t(), d),          # [9, C] tap-major
                    bn2=(_bf(sd[f"{q}.conv2.bn.weight"], d), _bf(sd[f"{q}.conv2.bn.bias"], d)),
                    se1=_bf(sd[f"{q}.se.fc1.weight"].flatten(1), d), se1b=_bf(sd[f"{q}.se.fc1.bias"], d),
                    se2=_bf(sd[f"{q}.se.fc2.weight"].flatten(1), d), se2b=_bf(sd[f"{q}.se.fc2.bias"], d),
                    conv3=_bf(sd[f"{q}.conv3.conv.weight"].flatten(1), d),
                    bn3=(_bf(sd[f"{q}.conv3.bn.weight"], d), _bf(sd[f"{q}.conv3.bn.bias"], d)))
                if f"{q}.downsample.conv.weight" in sd:
                    blk["ds"] = _bf(sd[f"{q}.downsample.conv.weight"].flatten(1), d)
                    blk["dsbn"] = (_bf(sd[f"{q}.downsample.bn.weight"], d), _bf(sd[f"{q}.downsample.bn.bias"], d))
                blocks.append(blk)
            return blocks
        self.s1 = stage(f"{M_PRE}.s1")
        self.s2 = stage(f"{M_PRE}.s2")
        sw = sd[f"{M_PRE}.sampler.0.weight"]                                        # [Co, Ci, 2,2,2]
        self.sampler_w = _bf(sw.permute(0, 2, 3, 4, 1).reshape(sw.shape[0], -1), d)  # [(kt,kh,kw,ci)] taps
        self.sampler_b = _bf(sd[f"{M_PRE}.sampler.0.bias"], d)
        self.ro0 = (_bf(sd[f"{M_PRE}.readout.0.weight"], d), _bf(sd[f"{M_PRE}.readout.0.bias"], d))
        self.ro2 = (_bf(sd[f"{M_PRE}.readout.2.weight"], d), _bf(sd[f"{M_PRE}.readout.2.bias"], d))
        # ---- decoder
        self.embed = _bf(sd["model.embed_tokens.weight"], d)
        self.layers = []
        for i in range(g.layers):
            p = f"model.layers.{i}"
            lw = dict(
                in_norm=_bf(sd[f"{p}.input_layernorm.weight"], d),
                post_norm=_bf(sd[f"{p}.post_attention_layernorm.weight"], d),
                wqkv=_bf(torch.cat([sd[f"{p}.self_attn.q_proj.weight"], sd[f"{p}.self_attn.k_proj.weight"],
                                    sd[f"{p}.self_attn.v_proj.weight"]], 0), d),
                wo=_bf(sd[f"{p}.self_attn.o_proj.weight"], d),
                wdown=_bf(sd[f"{p}.mlp.down_proj.weight"], d))
            # gate/up rows: interleaved in 16-row blocks when SwiGLU rides in the GEMM epilogue (frozen forward; LoRA forward,
            # whose epilogue also keeps the pre-activations - vlb_gemm_swiglu_save); plain [gate; up] when the weight itself
            # trains (full fine-tune: its gradient and optimiser state are in that layout)
            il = (not keep_transposed) if gate_up_interleaved is None else gate_up_interleaved
            wgu = _bf(torch.cat([sd[f"{p}.mlp.gate_proj.weight"], sd[f"{p}.mlp.up_proj.weight"]], 0), d)
            if keep_transposed:
                lw["wgu"] = wgu
                for k in ("wqkv", "wo", "wgu", "wdown"):
                    lw[k + "_t"] = ops.transpose(lw[k])
                if il:
                    del lw["wgu"]
            if il:
                lw["wgu_il"] = ops.interleave_gate_up(wgu[:g.ff], wgu[g.ff:])
            del wgu
            self.layers.append(lw)
        self.final_norm = _bf(sd["model.norm.weight"], d)
        inv = 1.0 / (g.rope_theta ** (torch.arange(0, g.head_dim, 2, dtype=torch.float32) / g.head_dim))
        fr = torch.arange(g.max_len, dtype=torch.float32)[:, None] * inv[None]
        self.rope_cos = fr.cos().to(d).contiguous()      # [S, D/2] fp32
        self.rope_sin = fr.sin().to(d).contiguous()

    @staticmethod
    def random_state_dict(g: Geometry, device, seed: int = 1234, dtype=BF16) -> dict:
        """Random-init weights of the real architecture directly on the device (no checkpoint is
        reachable offline): std-0.02 linears, unit norms, fan-in scaled connector convs."""
        gen = torch.Generator(device=device).manual_seed(seed)

        def rn(*shape, std=0.02):
            return (torch.randn(*shape, generator=gen, device=device, dtype=torch.float32) * std).to(dtype)

        def ones(n):
            return torch.ones(n, device=device, dtype=dtype)

        def zeros(n):
            return torch.zeros(n, device=device, dtype=dtype)
        sd = {}
        sd[f"{V_PRE}.embeddings.class_embedding"] = rn(g.vit_dim)
        sd[f"{V_PRE}.embeddings.patch_embedding.weight"] = rn(g.vit_dim, 3, g.patch, g.patch)
        sd[f"{V_PRE}.embeddings.position_embedding.weight"] = rn(g.grid * g.grid + 1, g.vit_dim)
        sd[f"{V_PRE}.pre_layrnorm.weight"] = ones(g.vit_dim)
        sd[f"{V_PRE}.pre_layrnorm.bias"] = zeros(g.vit_dim)
        for i in range(g.vit_layers):
            p = f"{V_PRE}.encoder.layers.{i}"
            for ln in ("layer_norm1", "layer_norm2"):
                sd[f"{p}.{ln}.weight"] = ones(g.vit_dim)
                sd[f"{p}.{ln}.bias"] = zeros(g.vit_dim)
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                sd[f"{p}.self_attn.{n}.weight"] = rn(g.vit_dim, g.vit_dim)
                sd[f"{p}.self_attn.{n}.bias"] = zeros(g.vit_dim)
            sd[f"{p}.mlp.fc1.weight"] = rn(g.vit_ff, g.vit_dim)
            sd[f"{p}.mlp.fc1.bias"] = zeros(g.vit_ff)
            sd[f"{p}.mlp.fc2.weight"] = rn(g.vit_dim, g.vit_ff)
            sd[f"{p}.mlp.fc2.bias"] = zeros(g.vit_dim)
        for st, cin in (("s1", g.vit_dim), ("s2", g.dim)):
            for b in range(g.proj_depth):
                ci = cin if b == 0 else g.dim
                q = f"{M_PRE}.{st}.b{b + 1}"
                rd = int(round(ci * g.proj_se_ratio))
                sd[f"{q}.conv1.conv.weight"] = rn(g.dim, ci, 1, 1, std=1 / math.sqrt(ci))
                sd[f"{q}.conv2.conv.weight"] = rn(g.dim, 1, 3, 3, std=1 / 3)
                sd[f"{q}.se.fc1.weight"] = rn(rd, g.dim, 1, 1, std=1 / math.sqrt(g.dim))
                sd[f"{q}.se.fc1.bias"] = zeros(rd)
                sd[f"{q}.se.fc2.weight"] = rn(g.dim, rd, 1, 1, std=1 / math.sqrt(rd))
                sd[f"{q}.se.fc2.bias"] = zeros(g.dim)
                sd[f"{q}.conv3.conv.weight"] = rn(g.dim, g.dim, 1, 1, std=1 / math.sqrt(g.dim))
                for n in ("conv1", "conv2", "conv3"):
                    sd[f"{q}.{n}.bn.weight"] = ones(g.dim)
                    sd[f"{q}.{n}.bn.bias"] = zeros(g.dim)
                if ci != g.dim:
                    sd[f"{q}.downsample.conv.weight"] = rn(g.dim, ci, 1, 1, std=1 / math.sqrt(ci))
                    sd[f"{q}.downsample.bn.weight"] = ones(g.dim)
                    sd[f"{q}.downsample.bn.bias"] = zeros(g.dim)
        sd[f"{M_PRE}.sampler.0.weight"] = rn(g.dim, g.dim, 2, 2, 2, std=1 / math.sqrt(8 * g.dim))
        sd[f"{M_PRE}.sampler.0.bias"] = zeros(g.dim)
        for n in ("readout.0", "readout.2"):
            sd[f"{M_PRE}.{n}.weight"] = rn(g.dim, g.dim, std=1 / math.sqrt(g.dim))
            sd[f"{M_PRE}.{n}.bias"] = zeros(g.dim)
        sd["model.embed_tokens.weight"] = rn(g.vocab, g.dim)
        for i in range(g.layers):
            p = f"model.layers.{i}"
            sd[f"{p}.self_attn.q_proj.weight"] = rn(g.heads * g.head_dim, g.dim)
            sd[f"{p}.self_attn.k_proj.weight"] = rn(g.kv_heads * g.head_dim, g.dim)
            sd[f"{p}.self_attn.v_proj.weight"] = rn(g.kv_heads * g.head_dim, g.dim)
            sd[f"{p}.self_attn.o_proj.weight"] = rn(g.dim, g.heads * g.head_dim)
            sd[f"{p}.mlp.gate_proj.weight"] = rn(g.ff, g.dim)
            sd[f"{p}.mlp.up_proj.weight"] = rn(g.ff, g.dim)
            sd[f"{p}.mlp.down_proj.weight"] = rn(g.dim, g.ff)
            sd[f"{p}.input_layernorm.weight"] = ones(g.dim)
            sd[f"{p}.post_attention_layernorm.weight"] = ones(g.dim)
        sd["model.norm.weight"] = ones(g.dim)
        return sd


class Backbone:
    """Forward of the frozen VideoLLaMA2 stack on libvlb kernels."""

    def __init__(self, g: Geometry, weights: Weights):
        self.g = g
        self.w = weights
        self.err_flag = torch.zeros(1, dtype=torch.int32, device=weights.dev)

    # ---------------- a4: CLIP tower (transformers modeling_clip.py:138-219,280-384)
    def vision_tower(self, vision_f32):
        """vision fp32 [N,3,H,W] -> patch features bf16 [N*grid^2, vit_dim] of hidden_states[-2]."""
        g, w = self.g, self.w
        N = vision_f32.shape[0]
        G, T = g.grid * g.grid, g.grid * g.grid + 1
        hd = g.vit_dim // g.vit_heads
        patches = ops.patchify(vision_f32, g.patch, g.patch_k_padded)
        pe = ops.gemm(patches, w.patch_w)
        x = ops.vit_assemble(pe, w.cls, w.pos, N, G, g.vit_dim)
        x = ops.layernorm(x, *w.pre_ln, g.vit_eps)
        D = g.vit_dim
        for lw in w.vit:
            h = ops.layernorm(x, *lw["ln1"], g.vit_eps)
            qkv = ops.gemm(h, lw["wqkv"], bias=lw["bqkv"])
            a = ops.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], N, T, g.vit_heads, g.vit_heads, hd,
                                  False, hd ** -0.5)
            x = ops.gemm(a, lw["wo"], bias=lw["bo"], residual=x)
            h = ops.layernorm(x, *lw["ln2"], g.vit_eps)
            h = ops.gemm(h, lw["w1"], bias=lw["b1"], act=ops.ACT_QUICK_GELU)
            x = ops.gemm(h, lw["w2"], bias=lw["b2"], residual=x)
        return ops.drop_cls(x, N, G, D)

    # ---------------- a5: STC connector (VideoLLaMA2 projector + timm RegStage)
    def _bottleneck(self, x, blk, N, H):
        g = self.g
        C = g.dim
        y = ops.gemm(x, blk["conv1"])
        y = ops.layernorm(y, *blk["bn1"], g.proj_eps, act=ops.ACT_SILU)
        y = ops.dwconv3x3(y, blk["dw"], N, H, H, C)
        y = ops.layernorm(y, *blk["bn2"], g.proj_eps, act=ops.ACT_SILU)
        s = ops.se_pool(y, N, H * H, C)
        s = ops.gemm(s, blk["se1"], bias=blk["se1b"], act=ops.ACT_SILU)
        s = ops.gemm(s, blk["se2"], bias=blk["se2b"])
        y = ops.se_scale(y, s, N, H * H, C, out=y)
        y = ops.gemm(y, blk["conv3"])
        sc = x
        if "ds" in blk:
            sc = ops.layernorm(ops.gemm(x, blk["ds"]), *blk["dsbn"], g.proj_eps)
        return ops.layernorm(y, *blk["bn3"], g.proj_eps, residual=sc, act=ops.ACT_SILU)

    def connector(self, feats, B):
        """feats bf16 [B*T*grid^2, vit_dim] -> video tokens bf16 [B*vis_tokens, dim]."""
        g, w = self.g, self.w
        x = feats
        for blk in w.s1:
            x = self._bottleneck(x, blk, B * g.num_frames, g.grid)
        cols = ops.im2col3d(x, B, g.num_frames, g.grid, g.grid, g.dim)
        x = ops.gemm(cols, w.sampler_w, bias=w.sampler_b, act=ops.ACT_SILU)
        for blk in w.s2:
            x = self._bottleneck(x, blk, B * g.ds_frames, g.ds_grid)
        x = ops.gemm(x, w.ro0[0], bias=w.ro0[1], act=ops.ACT_GELU)
        return ops.gemm(x, w.ro2[0], bias=w.ro2[1])

    # ---------------- a7: one Mistral decoder layer (modeling_mistral.py:202-240)
    def decoder_layer(self, x, lw, key_mask, B, S, save=None, layout=None):
        g = self.g
        qd, kd = g.heads * g.head_dim, g.kv_heads * g.head_dim
        h = ops.rmsnorm(x, lw["in_norm"], g.rms_eps)
        qkv = ops.gemm(h, lw["wqkv"])
        ops.rope_(qkv, self.w.rope_cos, self.w.rope_sin, B, S, g.heads + g.kv_heads, g.head_dim,
                  pos=None if layout is None else layout.pos)
        a = ops.attention_fwd(qkv[:, :qd], qkv[:, qd:qd + kd], qkv[:, qd + kd:], B, S, g.heads, g.kv_heads, g.head_dim,
                              True, g.head_dim ** -0.5, key_mask=key_mask, layout=layout)
        x = ops.gemm(a, lw["wo"], residual=x)
        h = ops.rmsnorm(x, lw["post_norm"], g.rms_eps)
        if lw.get("wgu_il") is not None:
            h = ops.gemm(h, lw["wgu_il"], act=ops.ACT_SWIGLU_PAIR)
        else:
            h = ops.swiglu(ops.gemm(h, lw["wgu"]))
        return ops.gemm(h, lw["wdown"], residual=x)

    def named_modules(self):
        """(upstream module name, leaf) for every ``nn.Linear`` of ``Videollama2MistralForCausalLM`` as the weights
        here imply it - what ``find_all_linear_names`` (reference litmodule :36-55) walks.  Convolutions of the
        connector / patch embedding are not linears upstream and are not listed; ``lm_head`` exists upstream
        (its logits are never read by the loss, so no weight is held here) and is listed so that the caller's
        explicit removal of it is exercised."""
        from .litmodule import _LinearInfo
        g = self.g
        qd, kd = g.heads * g.head_dim, g.kv_heads * g.head_dim
        for i in range(g.vit_layers):
            p = f"{V_PRE}.encoder.layers.{i}"
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                yield f"{p}.self_attn.{n}", _LinearInfo(g.vit_dim, g.vit_dim)
            yield f"{p}.mlp.fc1", _LinearInfo(g.vit_ff, g.vit_dim)
            yield f"{p}.mlp.fc2", _LinearInfo(g.vit_dim, g.vit_ff)
        for n in ("readout.0", "readout.2"):
            yield f"{M_PRE}.{n}", _LinearInfo(g.dim, g.dim)
        for i in range(g.layers):
            p = f"model.layers.{i}"
            yield f"{p}.self_attn.q_proj", _LinearInfo(qd, g.dim)
            yield f"{p}.self_attn.k_proj", _LinearInfo(kd, g.dim)
            yield f"{p}.self_attn.v_proj", _LinearInfo(kd, g.dim)
            yield f"{p}.self_attn.o_proj", _LinearInfo(g.dim, qd)
            yield f"{p}.mlp.gate_proj", _LinearInfo(g.ff, g.dim)
            yield f"{p}.mlp.up_proj", _LinearInfo(g.ff, g.dim)
            yield f"{p}.mlp.down_proj", _LinearInfo(g.dim, g.ff)
        yield "lm_head", _LinearInfo(g.vocab, g.dim)

    def splice(self, ids, video_tokens, layout=None):
        g = self.g
        return ops.splice_embed(ids, self.w.embed, video_tokens, g.vis_tokens, VIDEO_TOKEN_ID, self.err_flag, layout)

    def row_layout(self, ids_host=None, padvals=None):
        """Packed RowLayout for a batch whose ids (and optionally padvals) are still on the HOST, so the
        clip lengths cost no device sync: len[b] = (last non-pad id) + Nv, i.e. everything the reference's
        attention_mask (ids != 0, litmodule :271) keeps up to the right-padded tail.  Tokens past len[b]
        are attended by nobody and carry zero HRF weight, so dropping their rows changes no result.
        Returns None (dense layout) when the ids already live on the device."""
        g = self.g
        if ids_host is None or ids_host.device.type != "cpu":
            return None
        B, L = ids_host.shape
        S = L - 1 + g.vis_tokens
        nz = ids_host != 0
        last = L - 1 - torch.flip(nz, dims=[1]).to(torch.int8).argmax(dim=1)        # index of the last non-pad id
        lens = (last + g.vis_tokens).clamp(max=S)                                    # + (Nv - 1) + 1
        lens = torch.where(nz.any(dim=1), lens, torch.full_like(lens, S))
        if padvals is not None:               # never drop a row the weight mask could still weight
            lens = torch.maximum(lens, S - padvals[:, 0].to(lens.dtype).cpu().clamp(min=0))
        return ops.RowLayout(B, S, lens.tolist(), device=self.w.dev)

    # ---------------- fsdp.yaml-equivalent sharding of the frozen decoder weights (opt-in)
    def enable_sharding(self, group=None, comm=None):
        """Keep only this rank's 1/world shard of every decoder layer; full layers are all-gathered one
        layer ahead of compute on a side stream (parallel.ShardedLayerStore).  Frozen weights only: the full
        fine-tune's decoder weights are optimiser state (parallel.ShardedFlatState shards their gradients and moments)."""
        from .parallel import ShardedLayerStore
        if getattr(self, "trainable_decoder", False):
            raise ValueError("enable_sharding(): the decoder weights train in this configuration (full fine-tune); "
                             "only frozen decoder weights can be sharded per layer")
        side = torch.cuda.Stream(device=self.w.dev)
        keys = tuple(k for k in ("wqkv", "wo", "wgu", "wgu_il", "wdown") if k in self.w.layers[0])
        self.store = ShardedLayerStore(self.w.layers, keys, group, stream=side, comm=comm)
        self.store_t = None
        if "wqkv_t" in self.w.layers[0]:
            tkeys = ("wqkv_t", "wo_t", "wgu_t", "wdown_t")
            self.store_t = ShardedLayerStore(self.w.layers, tkeys, group, stream=side, comm=comm)
        for lw in self.w.layers:
            for k in list(lw):
                if k in keys or k.endswith("_t"):
                    lw[k] = None
        torch.cuda.empty_cache()

    def layer_weights(self, i, transposed=False, direction=1):
        """Weights of decoder layer i (gathered when sharded) and prefetch of the next one."""
        lw = self.w.layers[i]
        store = getattr(self, "store_t" if transposed else "store", None)
        if store is None:
            return lw
        full = store.get(i)
        store.prefetch(i + direction)
        return {**lw, **full}

    def decoder(self, x, key_mask, B, S, layer_outputs=None, layout=None):
        for i in range(len(self.w.layers)):
            x = self.decoder_layer(x, self.layer_weights(i), key_mask, B, S, layout=layout)
            if layer_outputs is not None:
                layer_outputs.append(x)
        return ops.rmsnorm(x, self.w.final_norm, self.g.rms_eps)

    # ---------------- frozen vision side one step ahead (software pipelining across steps)
    # The CLIP tower and the STC connector are frozen in the frozen-backbone and LoRA configurations (reference :86-99), so their
    # output for a batch depends on nothing the optimiser touches: the caller may start them for batch i+1 on a side stream
    # while step i's decoder runs (DevicePrefetcher does, bench.py does), and the step that consumes batch i+1 picks the result
    # up instead of computing it in line.  Every step still computes the tower once for one batch - nothing is cached or reused.
    #
    # A staged batch is identified by the IDENTITY of its pixel tensor, never by its address: every queue entry holds a
    # reference to the tensor it was computed from (which also pins the storage, so the caching allocator cannot hand the
    # address to a later batch) and lookups compare with ``is`` plus the version counter (an in-place edit invalidates).
    # Entries a consumer never asks for (rank-strided validation, limit_val_batches, a mid-epoch resume skip, max_steps) are
    # dropped by discard_video_tokens(): DevicePrefetcher calls it for every batch it staged but did not hand over.
    MAX_AHEAD = 4

    def _vision_compute(self, vision_f32, tower_only):
        g, B = self.g, vision_f32.shape[0]
        feats = self.vision_tower(vision_f32.reshape(B * g.num_frames, 3, g.image_size, g.image_size))
        return feats if tower_only else self.connector(feats, B)

    def prefetch_video_tokens(self, vision_f32, ready_event=None, tower_only=False):
        """Enqueue connector(vision_tower(pixels)) - or, ``tower_only`` (full fine-tune: the connector trains), the tower's
        output alone - for this batch on the side stream.  ``ready_event``: recorded after the pixels landed in HBM
        (DevicePrefetcher's copy); without it the side stream waits for the current stream."""
        dev = self.w.dev
        if getattr(self, "_vis_stream", None) is None:
            self._vis_stream = torch.cuda.Stream(device=dev)
        if getattr(self, "_vis_queue", None) is None:
            self._vis_queue = []
        st = self._vis_stream
        if ready_event is not None:
            st.wait_event(ready_event)
        else:
            st.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(st):
            vid = self._vision_compute(vision_f32, tower_only)
            ev = torch.cuda.Event()
            ev.record(st)
        vision_f32.record_stream(st)
        self._vis_queue.append((vision_f32, vision_f32._version, tower_only, vid, ev))
        del self._vis_queue[:-self.MAX_AHEAD]         # never more than a few batches ahead

    def defer_video_tokens(self, vision_f32, ready_event=None, tower_only=False):
        """Register a future batch; launch_deferred_video_tokens() (called by the step functions right behind their forward
        pass) enqueues it on the side stream."""
        if getattr(self, "_vis_pending", None) is None:
            self._vis_pending = []
        self._vis_pending.append((vision_f32, vision_f32._version, tower_only, ready_event))
        del self._vis_pending[:-self.MAX_AHEAD]

    def launch_deferred_video_tokens(self):
        pend, self._vis_pending = getattr(self, "_vis_pending", None) or [], []
        for vis, ver, tower_only, ev in pend:
            if vis._version == ver:                   # still the pixels that were registered
                self.prefetch_video_tokens(vis, ev, tower_only=tower_only)

    def discard_video_tokens(self, vision_f32=None):
        """Forget what was registered / computed ahead for this pixel tensor (``None``: for every batch): the batch will not be
        consumed.  Nothing is synchronised; a result still being computed on the side stream is simply released."""
        for name in ("_vis_queue", "_vis_pending"):
            q = getattr(self, name, None)
            if q:
                q[:] = [e for e in q if vision_f32 is not None and e[0] is not vision_f32]

    def video_tokens(self, vision_f32, tower_only=False):
        """connector(vision_tower(pixels)) [B, Nv, dim] (``tower_only``: the tower's tokens) - computed here, or taken from
        prefetch_video_tokens() when that ran for this very tensor object (and nobody wrote to it since).  A result that
        was prefetched ``tower_only`` is completed with the connector in line when the consumer wants the tokens (full
        fine-tune validation: the eval forward takes the frozen path)."""
        q = getattr(self, "_vis_queue", None)
        cur = torch.cuda.current_stream(self.w.dev)
        if q:
            for i, (vis, ver, t_only, vid, ev) in enumerate(q):
                if vis is not vision_f32:
                    continue
                del q[i]
                if ver != vision_f32._version or (t_only is False and tower_only):
                    break                             # stale pixels, or the tokens where the tower's features are wanted
                cur.wait_event(ev)
                vid.record_stream(cur)
                if t_only and not tower_only:
                    return self.connector(vid, vision_f32.shape[0])
                return vid
        pend = getattr(self, "_vis_pending", None)
        if pend:                                      # registered but its own step came first: compute it here, once
            for i, (vis, ver, t_only, ev) in enumerate(pend):
                if vis is vision_f32:
                    del pend[i]
                    if ev is not None:
                        cur.wait_event(ev)
                    break
        return self._vision_compute(vision_f32, tower_only)

    def forward(self, vision_f32, ids, stages=None, layout=None):
        """vision fp32 [B,T,3,H,W], ids int64 [B,L] -> hidden bf16 [rows, dim], key_mask uint8.
        rows = B*S (mask [B,S]) or, with a packed ``layout``, the clips' unpadded tokens (mask [rows])."""
        g = self.g
        B = vision_f32.shape[0]
        if stages is None:
            feats, vid = None, self.video_tokens(vision_f32)
        else:
            pix = vision_f32.reshape(B * g.num_frames, 3, g.image_size, g.image_size)
            feats = self.vision_tower(pix)
            vid = self.connector(feats, B)
        emb, key_mask = self.splice(ids, vid, layout)
        louts = [] if stages is not None else None
        hidden = self.decoder(emb, key_mask, B, g.max_len, louts, layout)
        if stages is not None:
            stages.update(vit_tokens=feats, video_tokens=vid, inputs_embeds=emb, key_mask=key_mask,
                          layer_outputs=louts, hidden=hidden)
        return hidden, key_mask
