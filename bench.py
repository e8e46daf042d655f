"""Headline benchmark: clips/sec of one fine-tuning step (BASELINE.json metric).

  python bench.py --gpus 1 --steps 8 --warmup 2            # configs[2]: 7B + LoRA r=16 + 2k head, B=3/GPU (the metric's config)
  python bench.py --workload frozen                        # configs[1]: 7B frozen backbone + 2k head, B=5/GPU
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" = forward through CLIP tower + STC connector + splice + 32 Mistral layers (+LoRA) + brain head,
backward through head and decoder (LoRA A/B gradients), global-norm clip, AdamW, cosine LR - all on
libvlb HIP kernels, inputs resident in HBM.  One JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel:
the 256x256 MFMA GEMM on the gate/up projection, timed with HIP events on its own stream inside the
timed region) and `cpu_baseline` (the oracle on the host cores, bounded sample, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (guides/MI355X_MICROARCH.md, chip-level parameters)
# L2->fabric bytes of ONE gate/up GEMM launch at B=5, measured offline with rocprofv3 --pmc (separate FETCH_SIZE /
# WRITE_SIZE passes, gfx950 2x read correction): profiles/r01_gemm_gateup_hbm_traffic.csv.  Only valid for that shape.
GATEUP_TRAFFIC_BYTES = {(5861, 28672, 4096): 2.495e9,     # default (LoRA, packed rows): profiles/r01_gemm_gateup_hbm_traffic_lora.csv
                        (9447, 28672, 4096): 3.746e9,     # --workload frozen, packed rows: ..._frozen_w4.csv
                        (10240, 28672, 4096): 3.904e9}    # frozen --no-pack, earlier ping-pong kernel: ..._traffic.csv
# algorithmic TFLOP per clip
TFLOP_PER_CLIP = {"frozen": 36.53, "lora": 67.8}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="lora", choices=["frozen", "lora"],
                    help="lora = BASELINE configs[2], the configuration the metric names (default); frozen = configs[1]")
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU (default 5 frozen / 3 lora, the reference's)")
    ap.add_argument("--geometry", default="7b", choices=["7b", "mini"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pack", action="store_true",
                    help="keep the dense [B, max_len] token layout (compute the padded tail rows too)")
    ap.add_argument("--shard-frozen", action="store_true",
                    help="fsdp.yaml-equivalent: keep 1/N of every frozen decoder layer per rank, all-gather one layer ahead")
    return ap.parse_args()


def cpu_baseline(g7, num_target):
    """Oracle (oracle/vlb_oracle.py, kind 'port') on the host cores: ONE clip through 2 ViT layers,
    the full connector, 2 decoder layers and the full head in fp32; ViT/decoder times are scaled to
    23/32 layers.  Returns (clips_per_s, cores, sample description)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vlb_oracle as O
    import dataclasses
    cores = torch.get_num_threads()
    g = dataclasses.replace(O.geometry_7b(num_target=num_target), vit_layers=3, layers=2)
    p = O.init_params(g, seed=1)
    batch = O.synthetic_batch(g, 1, seed=1)
    with torch.no_grad():
        t0 = time.perf_counter()
        pix = batch["vision"].reshape(g.num_frames, 3, g.image_size, g.image_size)
        vit = O.clip_tower(p, pix, g).view(1, g.num_frames, -1, g.vit_dim)
        t1 = time.perf_counter()
        vid = O.stc_connector(p, vit, g)
        t2 = time.perf_counter()
        emb, km = O.splice_multimodal(p["model.embed_tokens.weight"], batch["language"].long(), vid)
        hid = O.mistral_decoder(p, emb, km, g)
        t3 = time.perf_counter()
        wm = O.make_weight_mask(batch["padvals"], batch["vis_weights"], batch["lang_weights"], g.lang_len, g.max_len, g.ds_grid ** 2)
        pred, l2, _ = O.brain_head(p, hid, wm, g)
        t4 = time.perf_counter()
    t_vit, t_conn, t_dec, t_head = t1 - t0, t2 - t1, t3 - t2, t4 - t3
    total = t_vit * (23 / 2) + t_conn + t_dec * (32 / 2) + t_head
    sample = (f"1 clip fp32 forward: 2 ViT layers {t_vit:.1f}s x23/2 + connector {t_conn:.1f}s + 2 decoder layers "
              f"{t_dec:.1f}s x32/2 + head {t_head:.2f}s = {total:.0f}s/clip (extrapolated, forward only)")
    return 1.0 / total, cores, sample


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    ndev = torch.cuda.device_count()
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(ndev, 1)      # wraps only when rehearsing N ranks on fewer GPUs
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # VLB_FORCE_DIST=1 runs the multi-process code path (RCCL init, flat-bucket all-reduce, barriers)
    # even at world size 1 - used to rehearse `--gpus N` on a one-GPU box.
    use_dist = world > 1 or os.environ.get("VLB_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        backend = os.environ.get("VLB_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    from phantom_vlb_amd import ops
    if os.environ.get("VLB_GEMM_VARIANT"):        # A/B of GEMM kernel selection on the real step (tuning hook)
        import ctypes
        from phantom_vlb_amd._lib import lib
        lib.vlb_gemm_set_variant.argtypes, lib.vlb_gemm_set_variant.restype = [ctypes.c_int, ctypes.c_int], None
        lib.vlb_gemm_set_variant(int(os.environ["VLB_GEMM_VARIANT"], 0), 0)
    from phantom_vlb_amd.litmodule import VLBLitModule, VLBLitModuleConfig
    from phantom_vlb_amd.synthetic import synthetic_batch

    lora = a.workload == "lora"
    B = a.batch or (3 if lora else 5)
    cfg = VLBLitModuleConfig(
        model_path="DAMO-NLP-SG/VideoLLaMA2-7B", freeze_backbone=not lora, use_lora=lora,
        lora_r=16 if lora else None, lora_alpha=32 if lora else None, lora_dropout=0.1 if lora else None,
        dropout_rate=0.1, num_target=2048 if a.geometry == "7b" else 128, l2_lambda=1e-3, lr=1e-4, betas=[0.9, 0.999],
        eps=1e-8, weight_decay=1e-2, lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000,
        geometry=a.geometry, pack_tokens=not a.no_pack)
    import warnings
    warnings.simplefilter("ignore")
    m = VLBLitModule(cfg)
    m.world_size, m.rank = world, rank
    m.configure_model()
    if a.shard_frozen:
        m.backbone.enable_sharding()
    opt, sch = m.configure_optimizers()
    opt, sch = opt[0], sch[0]["scheduler"]
    if use_dist:
        from phantom_vlb_amd.parallel import attach_data_parallel, sync_module_states
        red = attach_data_parallel(m, opt)
        red.force = True                      # all-reduce even at world size 1 (rehearsal)
        sync_module_states(m)                 # rank 0's trainables everywhere + derived layouts rebuilt
    g = m.geometry
    batch = synthetic_batch(g, B, seed=1234 + rank, device=dev)
    # pixels / targets / weights are resident in HBM; the ids and padvals (20 KB) stay on the host, as a
    # DataLoader hands them over, so the step can size its unpadded row layout without a device sync
    batch["language"], batch["padvals"] = batch["language"].cpu(), batch["padvals"].cpu()
    lay = m.backbone.row_layout(batch["language"], batch["padvals"]) if m.pack_tokens else None
    rows_run, rows_dense = (lay.rows if lay is not None else B * g.max_len), B * g.max_len

    def step():
        loss = m.training_step(batch)
        opt.step()
        sch.step()
        return loss

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    # ---- dominant-kernel timing: event pairs around every gate/up GEMM launch of the timed steps
    probe = ops.enable_gemm_probe(N=2 * g.ff, K=g.dim)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    ops.disable_gemm_probe()
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    kern_ms, launches, (pM, pN, pK) = probe.result()
    if rank == 0:
        clips = world * B * a.steps
        value = clips / dt
        flops_launch = 2.0 * pM * pN * pK
        achieved = flops_launch / (kern_ms * 1e-3) / 1e12 if kern_ms else 0.0
        out = {
            "metric": "clips/sec fine-tune VideoLLaMA2-7B+LoRA->2k-voxel head, 1/2/4/8 MI355X",
            "value": round(value, 4), "unit": "clips/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("configs[1]: VideoLLaMA2-7B frozen backbone + linear 2k-voxel head, bf16"
                                    if not lora else "configs[2]: VideoLLaMA2-7B + LoRA r=16 + 2k-voxel head, bf16")
                       if a.geometry == "7b" else "configs[0]-shaped mini model (debug)",
                       "clips_per_gpu": B, "global_batch": world * B, "seq_len": g.max_len, "frames": g.num_frames,
                       "num_target": cfg.num_target, "weights": "random-init", "parallelism": f"dp{world}" + ("+sharded-frozen-weights" if a.shard_frozen else ""),
                       "loss": round(float(loss), 6),
                       "hbm_peak_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
                       "token_rows": {"computed": rows_run, "padded_layout": rows_dense,
                                      "note": "padded tail rows (ids == 0) are not computed, like the reference's flash-attn unpadding; results identical"},
                       # executed FLOPs: the per-clip figure scaled by the rows actually run (conservative: the vision tower is not reduced)
                       "step_tflops_per_gpu": round(value / world * TFLOP_PER_CLIP[a.workload] * rows_run / rows_dense, 1),
                       "step_frac_of_mfma_peak": round(value / world * TFLOP_PER_CLIP[a.workload] * rows_run / rows_dense / PEAK_BF16_TFLOPS, 4)},
            "roofline": {"bound": "mfma", "kernel": "gemm_w4_kernel (256x256x64 tile, 4 waves x 128x128) + the split-K launches of its partial last wave "
                         f"on the gate/up projection [{pM}x{pK}]x[{pN}x{pK}]^T, per vlb_gemm_bf16 call", "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                         "traffic": GATEUP_TRAFFIC_BYTES.get((pM, pN, pK)),
                         "traffic_note": "L2->fabric bytes of the main launch from rocprofv3 PMC passes (profiles/r01_gemm_gateup_hbm_traffic*.csv); "
                         + (f"algorithmic {2.0 * (pM * (pK + 64) + pN * (pK + 64)) + 2.0 * pM * pN:.3e} (A | t, W | B, C [M,N] bf16)" if lora else
                            f"algorithmic {2.0 * (pM * pK + pN * pK) + 1.0 * pM * pN:.3e} (A + W + SwiGLU-fused C [M,N/2] bf16)"),
                         "flops_per_launch": flops_launch, "avg_launch_ms": round(kern_ms, 4), "launches_timed": launches},
        }
        if world == 1 and not a.no_cpu_baseline and a.geometry == "7b":
            v, cores, sample = cpu_baseline(g, cfg.num_target)
            if lora:      # the oracle sample is forward only; the LoRA step adds dgrad + attention/LoRA backward
                v = v * TFLOP_PER_CLIP["frozen"] / TFLOP_PER_CLIP["lora"]
                sample += f"; LoRA step scaled by algorithmic FLOPs {TFLOP_PER_CLIP['lora']}/{TFLOP_PER_CLIP['frozen']} (backward not timed)"
            out["cpu_baseline"] = {"value": round(v, 6), "unit": "clips/s", "cores": cores, "kind": "port", "sample": sample}
        print(json.dumps(out), flush=True)
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
