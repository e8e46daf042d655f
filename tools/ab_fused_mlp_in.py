"""Same-process A/B on the 7B LoRA step: gate/up + SwiGLU + saved pre-activations in one GEMM epilogue
(vlb_gemm_swiglu_save, interleaved weight rows) against the plain GEMM followed by the SwiGLU kernel.
The module is built fused; the plain [gate; up] weights are added next to the interleaved ones for the B arm."""
import os
import statistics
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from phantom_vlb_amd.litmodule import VLBLitModule, VLBLitModuleConfig
    from phantom_vlb_amd.synthetic import synthetic_batch
    dev = torch.device("cuda:0")
    cfg = VLBLitModuleConfig(
        model_path="DAMO-NLP-SG/VideoLLaMA2-7B", freeze_backbone=False, use_lora=True, lora_r=16, lora_alpha=32, lora_dropout=0.1,
        dropout_rate=0.1, num_target=2048, l2_lambda=1e-3, lr=1e-4, betas=[0.9, 0.999], eps=1e-8, weight_decay=1e-2,
        lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000, geometry="7b", pack_tokens=True)
    warnings.simplefilter("ignore")
    m = VLBLitModule(cfg)
    m.configure_model()
    opt, sch = m.configure_optimizers()
    opt, sch = opt[0], sch[0]["scheduler"]
    ff = m.geometry.ff
    for lw in m.backbone.w.layers:            # de-interleave: rows (b, 0, r) are gate, (b, 1, r) up
        il = lw["wgu_il"].view(ff // 16, 2, 16, -1)
        lw["wgu"] = torch.cat([il[:, 0].reshape(ff, -1), il[:, 1].reshape(ff, -1)], 0).contiguous()
    batch = synthetic_batch(m.geometry, 3, seed=1234, device=dev)
    batch["language"], batch["padvals"] = batch["language"].cpu(), batch["padvals"].cpu()

    def step():
        loss = m.training_step(batch)
        opt.step()
        sch.step()
        return loss

    def arm(fused):
        m.lora.gu_il = fused
        m.lora._jobs_key = None
        m.lora.refresh()

    times, loss = {True: [], False: []}, {}
    for fused in (True, False):
        arm(fused)
        step()
    for rnd in range(5):
        for fused in (True, False):
            arm(fused)
            step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                l = step()
            torch.cuda.synchronize()
            times[fused].append((time.perf_counter() - t0) / 4 * 1e3)
            loss[fused] = float(l)
    for fused in (True, False):
        print(f"fused gate/up+SwiGLU+save = {fused}: min {min(times[fused]):.2f}  median {statistics.median(times[fused]):.2f} ms/step  (loss {loss[fused]:.5f})", flush=True)


if __name__ == "__main__":
    main()
