"""In-process A/B of one Python-side switch on the full 7B LoRA step (same box, same weights, interleaved rounds).

    python tools/ab_lora_step.py phantom_vlb_amd.lora:FUSE_SWIGLU_BWD 1 0
    python tools/ab_lora_step.py phantom_vlb_amd.ops:split_k_tails True False

Each round times 4 steps per setting; prints ms/step per setting (min over rounds) and the last loss."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    target, values = sys.argv[1], [eval(v) for v in sys.argv[2:]]
    mod_name, attr = target.split(":")
    mod = importlib.import_module(mod_name)
    from phantom_vlb_amd.litmodule import VLBLitModule, VLBLitModuleConfig
    from phantom_vlb_amd.synthetic import synthetic_batch
    dev = torch.device("cuda:0")
    cfg = VLBLitModuleConfig(
        model_path="DAMO-NLP-SG/VideoLLaMA2-7B", freeze_backbone=False, use_lora=True, lora_r=16, lora_alpha=32, lora_dropout=0.1,
        dropout_rate=0.1, num_target=2048, l2_lambda=1e-3, lr=1e-4, betas=[0.9, 0.999], eps=1e-8, weight_decay=1e-2,
        lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000, geometry="7b", pack_tokens=True)
    import warnings
    warnings.simplefilter("ignore")
    m = VLBLitModule(cfg)
    m.configure_model()
    opt, sch = m.configure_optimizers()
    opt, sch = opt[0], sch[0]["scheduler"]
    batch = synthetic_batch(m.geometry, 3, seed=1234, device=dev)
    batch["language"], batch["padvals"] = batch["language"].cpu(), batch["padvals"].cpu()

    def step():
        loss = m.training_step(batch)
        opt.step()
        sch.step()
        return loss

    best, loss = {repr(v): 1e9 for v in values}, {}
    for v in values:
        setattr(mod, attr, v)
        step()
    for rnd in range(4):
        for v in values:
            setattr(mod, attr, v)
            step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                l = step()
            torch.cuda.synchronize()
            best[repr(v)] = min(best[repr(v)], (time.perf_counter() - t0) / 4 * 1e3)
            loss[repr(v)] = float(l)
    for v in values:
        print(f"{target} = {v!r}: {best[repr(v)]:.2f} ms/step  (loss {loss[repr(v)]:.5f})", flush=True)


if __name__ == "__main__":
    main()
