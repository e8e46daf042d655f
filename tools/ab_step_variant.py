"""In-process A/B of GEMM kernel-variant words (tools build) on a whole 7B step: same box, same weights, interleaved rounds.

    python tools/ab_step_variant.py lora 3 0xC03          # configs[2] step: product tile order (3) vs order 0 (bits 10-12 = order XOR 3)
    python tools/ab_step_variant.py frozen 3 0xC03        # configs[1] step

Runs on libvlb_tools.so (the only build with vlb_gemm_set_variant); prints ms/step per variant (min and median over rounds)."""
import ctypes
import os
import statistics
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: E402,F401
from phantom_vlb_amd._lib import lib  # noqa: E402

lib.vlb_gemm_set_variant.argtypes = [ctypes.c_int, ctypes.c_int]
lib.vlb_gemm_set_variant.restype = None


def main():
    workload, variants = sys.argv[1], [int(v, 0) for v in sys.argv[2:]]
    from phantom_vlb_amd.litmodule import VLBLitModule, VLBLitModuleConfig
    from phantom_vlb_amd.synthetic import synthetic_batch
    dev = torch.device("cuda:0")
    lora = workload == "lora"
    cfg = VLBLitModuleConfig(
        model_path="DAMO-NLP-SG/VideoLLaMA2-7B", freeze_backbone=not lora, use_lora=lora, lora_r=16 if lora else None,
        lora_alpha=32 if lora else None, lora_dropout=0.1 if lora else None, dropout_rate=0.1, num_target=2048, l2_lambda=1e-3, lr=1e-4,
        betas=[0.9, 0.999], eps=1e-8, weight_decay=1e-2, lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000,
        geometry="7b", pack_tokens=True)
    warnings.simplefilter("ignore")
    m = VLBLitModule(cfg)
    m.configure_model()
    opt, sch = m.configure_optimizers()
    opt, sch = opt[0], sch[0]["scheduler"]
    batch = synthetic_batch(m.geometry, 3 if lora else 5, seed=1234, device=dev)
    batch["language"], batch["padvals"] = batch["language"].cpu(), batch["padvals"].cpu()

    def step():
        loss = m.training_step(batch)
        opt.step()
        sch.step()
        return loss

    times = {v: [] for v in variants}
    for v in variants:
        lib.vlb_gemm_set_variant(v, 0)
        step()
    for rnd in range(5):
        for v in variants:
            lib.vlb_gemm_set_variant(v, 0)
            step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                l = step()
            torch.cuda.synchronize()
            times[v].append((time.perf_counter() - t0) / 4 * 1e3)
    for v in variants:
        print(f"{workload} variant {v:#x}: min {min(times[v]):.2f}  median {statistics.median(times[v]):.2f} ms/step  (loss {float(l):.5f})", flush=True)
    lib.vlb_gemm_set_variant(3, 0)


if __name__ == "__main__":
    main()
