"""Shape constants of the VideoLLaMA2 -> brain-head path, parameterised.

The reference hard-codes them (NUM_FRAMES = 12, 13*13, 2048; src/litmodule/videollama2_vlb_litmodule.py:33,
180-181, 189-192); here they derive from one dataclass so the 8-frame mini configuration of
BASELINE.json configs[0] runs through the same kernels.
"""
from __future__ import annotations

from dataclasses import dataclass

VIDEO_TOKEN_ID = -201  # src/preprocessing/videollama2_vlb_extractfeatures.py:235-236


@dataclass
class Geometry:
    num_frames: int = 12
    image_size: int = 336
    patch: int = 14
    vit_dim: int = 1024
    vit_layers: int = 24
    vit_heads: int = 16
    vit_ff: int = 4096
    vit_eps: float = 1e-5
    vit_select_layer: int = -2
    proj_depth: int = 4
    proj_eps: float = 1e-6
    proj_se_ratio: float = 0.25
    dim: int = 4096
    layers: int = 32
    heads: int = 32
    kv_heads: int = 8
    head_dim: int = 128
    ff: int = 14336
    vocab: int = 32000
    rms_eps: float = 1e-5
    rope_theta: float = 1e6
    max_len: int = 2048
    num_target: int = 2048
    ln_eps: float = 1e-5
    l2_lambda: float = 1e-3
    lora_r: int = 16
    lora_alpha: int = 32

    @property
    def grid(self) -> int:
        return self.image_size // self.patch

    @property
    def ds_frames(self) -> int:
        return self.num_frames // 2 + 1

    @property
    def ds_grid(self) -> int:
        return self.grid // 2 + 1

    @property
    def vis_tokens(self) -> int:
        return self.ds_frames * self.ds_grid * self.ds_grid

    @property
    def lang_len(self) -> int:
        return self.max_len - self.vis_tokens + 1

    @property
    def vit_layers_run(self) -> int:
        return self.vit_layers + 1 + self.vit_select_layer

    @property
    def patch_k(self) -> int:
        return 3 * self.patch * self.patch

    @property
    def patch_k_padded(self) -> int:
        return (self.patch_k + 63) // 64 * 64


def geometry_7b(**kw) -> Geometry:
    """VideoLLaMA2-7B (Mistral-7B-Instruct + CLIP ViT-L/14-336 + STC connector), 12-frame clips."""
    return Geometry(**kw)


def geometry_mini(**kw) -> Geometry:
    """BASELINE.json configs[0]: 2-layer mini model, 128-voxel head, 8-frame clips.  Head widths are
    the production ones (ViT 64, decoder 128, GQA) so the same kernel instantiations run."""
    g = dict(num_frames=8, image_size=84, patch=14, vit_dim=128, vit_layers=3, vit_heads=2, vit_ff=256,
             dim=512, layers=2, heads=4, kv_heads=1, head_dim=128, ff=1024, vocab=512, max_len=128, num_target=128)
    g.update(kw)
    return Geometry(**g)


LORA_TARGETS = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
                "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")
