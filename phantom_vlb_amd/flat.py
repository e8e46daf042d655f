"""Flat storage for everything the optimiser touches.

All trainable masters (fp32), their bf16 compute copies, their gradients and the Adam moments live
in five flat buffers with one shared element order, so one step is ONE sum-of-squares launch, ONE
fused clip+AdamW launch (which also rewrites the bf16 copies) and - under data parallelism - ONE
all-reduce of the gradient buffer.  The per-tensor objects the rest of the code uses
(``head.master[n]``, ``lora.grads[n]``, the stacked LoRA ``A`` matrices ...) become views into them;
kernels always take ``.data_ptr()`` at call time, so re-pointing is transparent.
"""
from __future__ import annotations

import torch

BF16 = torch.bfloat16
# Segment ends (the head, every decoder layer) are padded to a multiple of 8 * lcm(1..8) elements, so any run of
# whole segments splits evenly into 1/world shards of 16-byte-aligned tensors for every world size up to 8 (and
# any other divisor of 840): the unit parallel.ShardedFlatState reduce-scatters and all-gathers.
SEG_ALIGN = 8 * 840


class FlatTrainables:
    def __init__(self, module):
        head, lora = module.head, getattr(module, "lora", None)
        dev = head.dev
        # ---- element order: head tensors, then per layer / group: A of each target (contiguous = the
        # stacked [R,in] matrix), then B^T of each target
        entries = []                          # (name, numel, shape)
        for n in head.master:
            entries.append((n, head.master[n].numel(), tuple(head.master[n].shape)))
        groups = []
        self.layer_ranges = []                # per decoder layer: [start, end) of its LoRA tensors in the flat order
        if lora is not None:
            from .lora import GROUPS
            for li in range(len(lora.layers)):
                for gname, targets in GROUPS:
                    a_names = [f"model.layers.{li}.{t}.lora_A.weight" for t in targets]
                    b_names = [f"model.layers.{li}.{t}.lora_B.weight" for t in targets]
                    groups.append((li, gname, a_names, b_names))
                    for n in a_names + b_names:
                        entries.append((n, lora.master[n].numel(), tuple(lora.master[n].shape)))
        def seg_of(n):
            return int(n.split(".")[2]) if n.startswith("model.layers.") else -1      # -1 = the head

        offs, off, seg_start = {}, 0, 0
        for i, (n, k, shp) in enumerate(entries):
            offs[n] = (off, k, shp)
            off += (k + 7) // 8 * 8              # keep every tensor 32-byte aligned (16 B in bf16)
            if i + 1 == len(entries) or seg_of(entries[i + 1][0]) != seg_of(n):
                off = (off + SEG_ALIGN - 1) // SEG_ALIGN * SEG_ALIGN
                if seg_of(n) < 0:
                    self.head_range = (seg_start, off)
                else:
                    self.layer_ranges.append((seg_start, off))
                seg_start = off
        self.numel = off
        self.master = torch.zeros(off, dtype=torch.float32, device=dev)
        self.compute = torch.zeros(off, dtype=BF16, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.offsets = offs

        def view(buf, n):
            o, k, shp = offs[n]
            return buf[o:o + k].view(shp)

        for n in list(head.master):
            view(self.master, n).copy_(head.master[n])
            view(self.compute, n).copy_(head.compute[n])
            head.master[n] = view(self.master, n)
            head.compute[n] = view(self.compute, n)
            head.grads[n] = view(self.grad, n)
        if lora is not None:
            for li, gname, a_names, b_names in groups:
                blk = lora.layers[li][gname]
                o0 = offs[a_names[0]][0]
                R, kin = blk["A"].shape
                assert all(offs[a_names[j]][0] == o0 + j * lora.rp * kin for j in range(len(a_names)))
                for n in a_names + b_names:
                    view(self.master, n).copy_(lora.master[n])
                    lora.master[n] = view(self.master, n)
                    lora.grads[n] = view(self.grad, n)
                self.compute[o0:o0 + R * kin].view(R, kin).copy_(blk["A"])
                blk["A"] = self.compute[o0:o0 + R * kin].view(R, kin)
                lora.grad_A[li][gname] = self.grad[o0:o0 + R * kin].view(R, kin)
                for n in b_names:
                    view(self.compute, n).copy_(lora.bt[n])
                    lora.bt[n] = view(self.compute, n)
        self.names = [n for n, _, _ in entries]

    def named_masters(self):
        return [(n, self.master[o:o + k].view(shp)) for n, (o, k, shp) in self.offsets.items()]
