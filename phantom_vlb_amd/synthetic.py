"""Seeded synthetic clips in the lazy-load sample schema (SURVEY.md 8d / src/datamodule/...:98-109).

Used by bench.py, smoke() and the built-in runner when no lazy-load files are given: there is no
network for CNeuroMod data.  Shapes and dtypes are what ``VLB_Dataset.__getitem__`` + default
collate hand to ``training_step``: vision fp32 [B,T,3,H,W], language fp32 ids [B,L] with one -201
slot and a right-padded tail, timeseries fp32 [B,V], padvals int64 [B,3], vis/lang weights f64.
"""
from __future__ import annotations

import torch

from .geometry import Geometry, VIDEO_TOKEN_ID

# Glover-HRF samples at 1.49*[5.5,5.0,...,2.5] s (lazyloading.py:108-115); any fixed positive vector works
GLOVER_VIS_WEIGHTS_7 = (0.0463, 0.0644, 0.0829, 0.0964, 0.0982, 0.0826, 0.0491)


def synthetic_batch(g: Geometry, batch: int, seed: int = 1234, device="cpu", inst_len: int = 9) -> dict:
    gen = torch.Generator().manual_seed(seed)
    L = g.lang_len
    language = torch.zeros(batch, L)
    padvals = torch.zeros(batch, 3, dtype=torch.int64)
    base = torch.tensor(GLOVER_VIS_WEIGHTS_7, dtype=torch.float64)
    vis_w = torch.stack([base[torch.arange(g.ds_frames) % 7] * (1 + 0.05 * b) for b in range(batch)])
    lang_w = torch.zeros(batch, 64, dtype=torch.float64)
    max_dialog = min(58, L - (2 + inst_len + 4) - 2)
    for b in range(batch):
        dialog_len = int(torch.randint(0, max_dialog + 1, (1,), generator=gen))
        body = 2 + inst_len + dialog_len + 4
        pad_len = int(torch.randint(0, min(300, L - 1 - body - 1) + 1, (1,), generator=gen))
        P = L - 1 - body - pad_len
        ids = torch.randint(3, g.vocab, (L,), generator=gen).float()
        ids[P] = VIDEO_TOKEN_ID
        if pad_len:
            ids[L - pad_len:] = 0
        language[b] = ids
        padvals[b] = torch.tensor([pad_len, inst_len, dialog_len])
        lang_w[b, :dialog_len] = torch.rand(dialog_len, generator=gen, dtype=torch.float64) * 0.2
    dev = torch.device(device)
    if dev.type == "cuda":   # 16 MB/clip of pixels: generate on the device
        dgen = torch.Generator(device=dev).manual_seed(seed)
        vision = torch.randn(batch, g.num_frames, 3, g.image_size, g.image_size, generator=dgen, device=dev)
        timeseries = torch.randn(batch, g.num_target, generator=dgen, device=dev)
    else:
        vision = torch.randn(batch, g.num_frames, 3, g.image_size, g.image_size, generator=gen)
        timeseries = torch.randn(batch, g.num_target, generator=gen)
    out = dict(vision=vision, language=language, timeseries=timeseries, padvals=padvals, vis_weights=vis_w,
               lang_weights=lang_w)
    return {k: v.to(dev) for k, v in out.items()}
