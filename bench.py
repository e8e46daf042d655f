"""Headline benchmark: clips/sec of one fine-tuning step (BASELINE.json metric).

  python bench.py --gpus 1 --steps 8 --warmup 2            # configs[2]: 7B + LoRA r=16 + 2k head, B=3/GPU (the metric's config)
  python bench.py --workload frozen                        # configs[1]: 7B frozen backbone + 2k head, B=5/GPU
  python bench.py --gpus N                                 # launches N ranks itself (one per GPU, RCCL), prints rank 0's line
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...     # what the driver does; same result
  python bench.py --gpus N --shard-frozen                  # fsdp.yaml-equivalent: frozen decoder weights sharded 1/N as well

A "step" = forward through CLIP tower + STC connector + splice + 32 Mistral layers (+LoRA) + brain head,
backward through head and decoder (LoRA A/B gradients), global-norm clip, AdamW, cosine LR - all on
libvlb HIP kernels, inputs resident in HBM.  One JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel:
the 256x256 MFMA GEMM call of the gate/up projection, timed with HIP events on its own stream inside the
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
PEAK_FP8_TFLOPS = 5000.0       # dense fp8 / MX-fp8 MFMA (same source; the headline figures with 2:1 sparsity are never used)


def gateup_traffic(shape, prefix=""):
    """L2->fabric bytes of ONE gate/up GEMM call of this (M, N, K), as measured with rocprofv3 --pmc (separate FETCH_SIZE /
    WRITE_SIZE passes, gfx950 2x read correction) and recorded by `tools/profile_tables.py traffic` in the tracked file
    profiles/gateup_traffic.json - the line and profiles/ cannot disagree.  (bytes, source file) or (None, None)."""
    try:
        with open(os.path.join(ROOT, "profiles", "gateup_traffic.json")) as f:
            e = json.load(f).get(prefix + "%d,%d,%d" % tuple(shape))
        return (float(e["bytes"]), e["source"]) if e else (None, None)
    except (OSError, ValueError, KeyError):
        return None, None


# algorithmic TFLOP per clip
TFLOP_PER_CLIP = {"frozen": 36.53, "lora": 67.8, "full": 100.8}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="lora", choices=["frozen", "lora", "full"],
                    help="lora = BASELINE configs[2], the configuration the metric names (default); frozen = configs[1]; "
                         "full = configs[4]'s model side on the GPUs given: full-parameter fine-tune, 65k-voxel head")
    ap.add_argument("--fp8", action="store_true", help="--workload full: decoder forward / dgrad GEMMs on the MX-fp8 MFMA path")
    ap.add_argument("--num-target", type=int, default=0, help="head width (default 2048; 65536 for --workload full)")
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU (default 5 frozen / 3 lora, the reference's)")
    ap.add_argument("--geometry", default="7b", choices=["7b", "mini"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pack", action="store_true",
                    help="keep the dense [B, max_len] token layout (compute the padded tail rows too)")
    ap.add_argument("--shard-frozen", action="store_true",
                    help="fsdp.yaml-equivalent: keep 1/N of every frozen decoder layer per rank, all-gather one layer ahead")
    return ap.parse_args()


def cpu_baseline(num_target, lora):
    """Oracle (oracle/vlb_oracle.py, kind 'port') on the host cores, bounded sample: ONE clip through 2 ViT layers,
    the full connector, 2 decoder layers and the full head in fp32 - forward, and for the LoRA workload ALSO the
    backward pass (torch autograd through head + adapted decoder layers, exactly the gradients the step needs);
    ViT / decoder times are scaled to 23 / 32 layers.  Plus configs[0] exactly: one full training
    step (forward, backward, clip, AdamW) of the mini model on 4 clips.  Returns the cpu_baseline object."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vlb_oracle as O
    import dataclasses
    cores = torch.get_num_threads()
    kw = dict(lora_r=16, lora_alpha=32) if lora else {}
    g = dataclasses.replace(O.geometry_7b(num_target=num_target, **kw), vit_layers=3, layers=2)
    p = O.init_params(g, seed=1, lora=lora, lora_b_std=0.02) if lora else O.init_params(g, seed=1)
    train = O.trainable_names(p, freeze_backbone=not lora, use_lora=lora)
    for n in train:
        p[n].requires_grad_(True)
    batch = O.synthetic_batch(g, 1, seed=1)
    t0 = time.perf_counter()
    with torch.no_grad():                                  # frozen in every measured configuration: forward only
        pix = batch["vision"].reshape(g.num_frames, 3, g.image_size, g.image_size)
        vit = O.clip_tower(p, pix, g).view(1, g.num_frames, -1, g.vit_dim)
        t1 = time.perf_counter()
        vid = O.stc_connector(p, vit, g)
        t2 = time.perf_counter()
        emb, km = O.splice_multimodal(p["model.embed_tokens.weight"], batch["language"].long(), vid)
    with torch.set_grad_enabled(lora):
        hid = O.mistral_decoder(p, emb, km, g)
    t3 = time.perf_counter()
    wm = O.make_weight_mask(batch["padvals"], batch["vis_weights"], batch["lang_weights"], g.lang_len, g.max_len, g.ds_grid ** 2)
    hid_head = hid if lora else hid.detach()
    pred, l2, _ = O.brain_head(p, hid_head, wm, g)
    loss = torch.nn.functional.mse_loss(pred, batch["timeseries"]) + l2
    t4 = time.perf_counter()
    if lora:                                               # decoder backward dominates; head backward rides along
        loss.backward()
        t5 = time.perf_counter()
        t_bwd_dec, t_bwd_head = t5 - t4, 0.0
    else:
        loss.backward()                                    # head only
        t5 = time.perf_counter()
        t_bwd_dec, t_bwd_head = 0.0, t5 - t4
    t_vit, t_conn, t_dec, t_head = t1 - t0, t2 - t1, t3 - t2, t4 - t3
    total = t_vit * (23 / 2) + t_conn + (t_dec + t_bwd_dec) * (32 / 2) + t_head + t_bwd_head
    sample = (f"1 clip fp32, oracle on the host: 2 ViT layers fwd {t_vit:.1f}s x23/2 + connector fwd {t_conn:.1f}s + 2 decoder layers "
              f"fwd {t_dec:.1f}s" + (f" + bwd (autograd, LoRA+head grads) {t_bwd_dec:.1f}s" if lora else "")
              + f" x32/2 + head fwd {t_head:.2f}s" + (f" + bwd {t_bwd_head:.2f}s" if not lora else "")
              + f" = {total:.0f}s/clip (layers extrapolated; optimiser update not included)")
    # ---- configs[0], exact: mini model, 4 clips, one full training step on the CPU
    gm = O.geometry_mini(**kw)
    pm = O.round_bf16(O.init_params(gm, seed=1234, lora=lora, lora_b_std=0.02) if lora else O.init_params(gm, seed=1234))
    bm = O.synthetic_batch(gm, 4, seed=1234)
    names = O.trainable_names(pm, freeze_backbone=not lora, use_lora=lora)

    def mini_step():
        q = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in pm.items()}
        l, _ = O.training_loss(q, bm, gm)
        l.backward()
        grads, _ = O.clip_grad_norm({k: q[k].grad for k in names}, 1.0)
        for k in names:
            O.adamw_step(q[k].detach(), grads[k], torch.zeros_like(grads[k]), torch.zeros_like(grads[k]), 1, 1e-4)
    mini_step()
    tm = time.perf_counter()
    reps = 3
    for _ in range(reps):
        mini_step()
    tm = (time.perf_counter() - tm) / reps
    return {"value": round(1.0 / total, 6), "unit": "clips/s", "cores": cores, "kind": "port", "sample": sample,
            "mini": {"value": round(4 / tm, 3), "unit": "clips/s", "workload": "configs[0]: mini 2-layer model + 128-voxel head, 4 clips, "
                     "one full training step (fwd + bwd + clip + AdamW), exact (no extrapolation)", "ms_per_step": round(tm * 1e3, 2)}}


def launch_ranks(a) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) BEFORE this process touches the GPU,
    relay their output (rank 0 prints the JSON line) and return their exit status.  Never falls back to fewer ranks:
    with fewer than N visible devices this fails, unless VLB_DIST_BACKEND=gloo asks for a time-shared rehearsal."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()                  # does not initialise the GPU
    rehearsal = os.environ.get("VLB_DIST_BACKEND", "nccl") != "nccl"
    if ndev < a.gpus and not rehearsal:
        print(f"bench.py: --gpus {a.gpus} but only {ndev} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    ndev = torch.cuda.device_count()
    backend = os.environ.get("VLB_DIST_BACKEND", "nccl")
    if backend == "nccl" and world > ndev:
        print(f"bench.py: {world} RCCL ranks need {world} GPUs, {ndev} visible", file=sys.stderr)
        sys.exit(2)
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(ndev, 1)      # wraps only in a gloo rehearsal of N ranks on fewer GPUs
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # VLB_FORCE_DIST=1 runs the multi-process code path (RCCL init, reduce-scatter / all-gather of the sharded
    # optimiser state, barriers) even at world size 1 - used to rehearse `--gpus N` on a one-GPU box.
    force = os.environ.get("VLB_FORCE_DIST") == "1"
    use_dist = world > 1 or force
    if use_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    from phantom_vlb_amd import _lib, ops
    if not _lib.IS_PRODUCT_LIB:
        print(f"bench.py: refusing to benchmark {_lib.LIB_PATH} (VLB_LIB is set): only the in-tree product library "
              "phantom_vlb_amd/libvlb.so is measured", file=sys.stderr)
        sys.exit(2)
    from phantom_vlb_amd.litmodule import VLBLitModule, VLBLitModuleConfig
    from phantom_vlb_amd.synthetic import synthetic_batch

    lora, full = a.workload == "lora", a.workload == "full"
    if full and a.shard_frozen:
        print("bench.py: --shard-frozen shards FROZEN decoder weights; with --workload full they train (their gradients and "
              "optimiser state are sharded by the data-parallel step itself)", file=sys.stderr)
        sys.exit(2)
    B = a.batch or (5 if a.workload == "frozen" else 3)
    num_target = a.num_target or (65536 if full else 2048)
    cfg = VLBLitModuleConfig(
        model_path="DAMO-NLP-SG/VideoLLaMA2-7B", freeze_backbone=a.workload == "frozen", use_lora=lora,
        lora_r=16 if lora else None, lora_alpha=32 if lora else None, lora_dropout=0.1 if lora else None,
        dropout_rate=0.1, num_target=num_target if a.geometry == "7b" else 128, l2_lambda=1e-3, lr=1e-4, betas=[0.9, 0.999],
        eps=1e-8, weight_decay=1e-2, lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000,
        geometry=a.geometry, pack_tokens=not a.no_pack, fp8_gemm=bool(a.fp8 and full))
    import warnings
    warnings.simplefilter("ignore")
    m = VLBLitModule(cfg)
    m.world_size, m.rank = world, rank
    m.configure_model()
    if a.shard_frozen:
        m.backbone.enable_sharding()
    opt, sch = m.configure_optimizers()
    opt, sch = opt[0], sch[0]["scheduler"]
    comm_name = "none"
    if use_dist:
        from phantom_vlb_amd.parallel import attach_data_parallel, sync_module_states
        state = attach_data_parallel(m, opt, force_collectives=force)
        sync_module_states(m)                 # rank 0's trainables everywhere + derived layouts rebuilt
        comm_name = f"{type(state.comm).__name__}/{backend} world={torch.distributed.get_world_size()}"
    g = m.geometry
    batch = synthetic_batch(g, B, seed=1234 + rank, device=dev)
    # pixels / targets / weights are resident in HBM; the ids and padvals (20 KB) stay on the host, as a
    # DataLoader hands them over, so the step can size its unpadded row layout without a device sync
    batch["language"], batch["padvals"] = batch["language"].cpu(), batch["padvals"].cpu()
    lay = m.backbone.row_layout(batch["language"], batch["padvals"]) if m.pack_tokens else None
    rows_run, rows_dense = (lay.rows if lay is not None else B * g.max_len), B * g.max_len

    # Software pipelining across steps, as Trainer.fit's DevicePrefetcher does it: the frozen vision side (CLIP tower + STC
    # connector) of the NEXT step's batch is enqueued on a side stream before this step's decoder work and picked up by the
    # next training_step.  Every step computes it exactly once for one batch (nothing is cached: the queue entry is consumed);
    # with --workload full only the CLIP tower runs ahead (the connector trains).  VLB_BENCH_VISION_PREFETCH=0 keeps everything on one stream.
    pipelined = os.environ.get("VLB_BENCH_VISION_PREFETCH", "1") == "1"
    if pipelined:
        m.prefetch_vision(batch)              # the first step's features

    def step():
        if pipelined:
            m.prefetch_vision(batch)          # the next step's
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
    ops.profile_marker()                      # kernel traces are cut to [marker, marker] by tools/profile_tables.py
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    ops.profile_marker()
    ops.disable_gemm_probe()
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    kern_ms, launches, (pM, pN, pK) = probe.result()
    # ---- the whole long-K GEMM family (every four-wave launch: plain, gate/up + SwiGLU, masked-pair dgrads, with their tail
    # launches), from event pairs around each call in TWO EXTRA steps after the timed region (the pairs cost ~2 us per GEMM,
    # so they stay out of `value`); vision prefetch off for these so no side-stream kernel runs under a timed call
    fam_steps = 2 if a.geometry == "7b" and not a.fp8 else 0
    family = None
    if fam_steps and rank == 0 or (fam_steps and use_dist):
        fprobe = ops.enable_gemm_probe(N=2 * g.ff, K=g.dim, family=True)
        was = pipelined
        pipelined = False
        for _ in range(fam_steps):
            step()
        pipelined = was
        ops.disable_gemm_probe()
        fam = fprobe.family_result()
        tot_ms = sum(v[1] for v in fam.values())
        tot_tf = sum(v[2] for v in fam.values())
        family = {"what": "every bf16 GEMM call with K + K2 >= 4096 (the four-wave kernels incl. 192-row, masked-pair and split-K / re-cut "
                          "tail launches), HIP-event pairs around each call in %d extra untimed steps, one stream" % fam_steps,
                  "ms_per_step": round(tot_ms / fam_steps, 2), "tflop_per_step": round(tot_tf / fam_steps, 2),
                  "tflops": round(tot_tf / (tot_ms * 1e-3), 1) if tot_ms else 0.0,
                  "frac_of_bf16_peak": round(tot_tf / (tot_ms * 1e-3) / PEAK_BF16_TFLOPS, 4) if tot_ms else 0.0,
                  "by_kind": {k: {"calls_per_step": v[0] // fam_steps, "ms_per_step": round(v[1] / fam_steps, 2),
                                  "tflops": round(v[2] / (v[1] * 1e-3), 1)} for k, v in sorted(fam.items())}}
    # the recorded collections are per call KIND: LoRA / frozen calls under the bare shape, the fp8 call under "fp8:", the plain
    # bf16 call of the full fine-tune (no collection on record) under "full:"
    traffic, traffic_src = gateup_traffic((pM, pN, pK), "fp8:" if a.fp8 else "full:" if full else "")
    if rank == 0:
        clips = world * B * a.steps
        value = clips / dt
        flops_launch = 2.0 * pM * pN * pK
        achieved = flops_launch / (kern_ms * 1e-3) / 1e12 if kern_ms else 0.0
        peak_tf = PEAK_FP8_TFLOPS if a.fp8 else PEAK_BF16_TFLOPS
        out = {
            "metric": "clips/sec fine-tune VideoLLaMA2-7B+LoRA->2k-voxel head, 1/2/4/8 MI355X",
            "value": round(value, 4), "unit": "clips/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "mxfp8 (e4m3 + E8M0 block scales) decoder GEMMs, bf16 elsewhere" if a.fp8 else "bf16", "data": "synthetic",
            "config": {"workload": {"frozen": "configs[1]: VideoLLaMA2-7B frozen backbone + linear 2k-voxel head, bf16",
                                    "lora": "configs[2]: VideoLLaMA2-7B + LoRA r=16 + 2k-voxel head, bf16",
                                    "full": f"configs[4] model side: VideoLLaMA2-7B full-parameter fine-tune (all but the vision tower), "
                                            f"{num_target}-voxel head, " + ("MX-fp8 decoder forward / dgrad / wgrad GEMMs" if a.fp8 else "bf16 GEMMs")}[a.workload]
                       if a.geometry == "7b" else "configs[0]-shaped mini model (debug)",
                       "clips_per_gpu": B, "global_batch": world * B, "seq_len": g.max_len, "frames": g.num_frames,
                       "num_target": cfg.num_target, "weights": "random-init",
                       "parallelism": f"dp{world}: clips sharded over ranks; trainables' gradients reduce-scattered, fp32 master + Adam moments "
                                      f"sharded 1/{world}, bf16 copies all-gathered; frozen weights "
                                      + ("sharded 1/N per layer, all-gathered one layer ahead (fsdp.yaml FULL_SHARD equivalent)"
                                         if a.shard_frozen else "replicated (--shard-frozen for the fsdp.yaml-equivalent layout)")
                                      + ("" if not full else "; trained decoder weights " + (
                                          "FULL_SHARD (fsdp.yaml:11): 1/N of every layer per rank, a layer all-gathered for its forward and again for its "
                                          "backward, one layer ahead" if getattr(getattr(m, "sharded_backbone", None), "full_shard", False)
                                          else "replicated (SHARD_GRAD_OP / single rank)")),
                       "comm": comm_name, "sharded_frozen_variant": None,
                       "vision_prefetch": (("CLIP tower" if full else "CLIP tower + connector") + " of step i+1 run on a side stream under step i's backward pass (as Trainer.fit's "
                                           "DevicePrefetcher does); each step computes them once, nothing is reused") if pipelined else "off",
                       "loss": round(float(loss), 6),
                       "hbm_peak_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
                       "token_rows": {"computed": rows_run, "padded_layout": rows_dense,
                                      "note": "padded tail rows (ids == 0) are not computed, like the reference's flash-attn unpadding; results identical"},
                       # executed FLOPs: the per-clip figure scaled by the rows actually run (conservative: the vision tower is not reduced)
                       "step_tflops_per_gpu": round(value / world * TFLOP_PER_CLIP[a.workload] * rows_run / rows_dense, 1),
                       "step_frac_of_bf16_mfma_peak": round(value / world * TFLOP_PER_CLIP[a.workload] * rows_run / rows_dense / PEAK_BF16_TFLOPS, 4)},
            "roofline": {"bound": "mfma", "kernel": (f"gemm_mxfp8 kernel (256x256x128 tile, 4 waves x 128x128, v_mfma_scale_f32_16x16x128_f8f6f4) + the re-cut launch of its partial "
                                                     f"last round on the gate/up projection [{pM}x{pK}]x[{pN}x{pK}]^T, per vlb_gemm_mxfp8 call; peak = dense fp8 MFMA" if a.fp8 else
                                                     "gemm_w4_kernel (256x256x64 tile, 4 waves x 128x128) + the split-K launches of its partial last wave "
                         f"on the gate/up projection [{pM}x{pK}]x[{pN}x{pK}]^T, per "
                         + ("vlb_gemm_swiglu_save call (epilogue: + LoRA pair, SwiGLU, saved pre-activations; FLOPs counted: the base GEMM only)" if lora
                            else "vlb_gemm_bf16 call")), "achieved": round(achieved, 1), "peak": peak_tf,
                         "unit": "TFLOP/s", "frac": round(achieved / peak_tf, 4),
                         "traffic": traffic,
                         "traffic_note": (f"L2->fabric bytes per call from rocprofv3 PMC passes ({traffic_src}, via profiles/gateup_traffic.json); "
                                          if traffic is not None else "no PMC collection on record for this shape (profiles/gateup_traffic.json); ")
                         + (f"algorithmic {2.0 * (pM * (pK + 64) + pN * (pK + 64)) + 2.0 * pM * pN + 1.0 * pM * pN:.3e} (A | t, W | B, saved [gate|up] [M,N] and silu(gate)*up [M,N/2], bf16)" if lora else
                            f"algorithmic {(pM * pK + pN * pK) * (1 + 1 / 32) + 2.0 * pM * pN:.3e} (e4m3 A + W with their E8M0 scales, C [M,N] bf16)" if a.fp8 else
                            f"algorithmic {2.0 * (pM * pK + pN * pK) + 2.0 * pM * pN:.3e} (A + W + C [M,N] bf16)" if full else
                            f"algorithmic {2.0 * (pM * pK + pN * pK) + 1.0 * pM * pN:.3e} (A + W + SwiGLU-fused C [M,N/2] bf16)"),
                         "flops_per_launch": flops_launch, "avg_launch_ms": round(kern_ms, 4), "launches_timed": launches},
            "gemm_family": family,
        }
        if world == 1 and not a.no_cpu_baseline and a.geometry == "7b" and not full:
            out["cpu_baseline"] = cpu_baseline(cfg.num_target, lora)
    else:
        out = None

    import threading
    emit_lock, emitted = threading.Lock(), []

    def emit(variant):
        """Exactly one JSON line, whoever gets here first (the main thread or the watchdog)."""
        with emit_lock:
            if emitted:
                return
            emitted.append(True)
            if rank == 0:
                out["config"]["sharded_frozen_variant"] = variant
                print(json.dumps(out), flush=True)

    # ---- N > 1: the fsdp.yaml-equivalent layout of the frozen decoder next to the replicated default (SURVEY.md 8e asks
    # for both): after the headline measurement, keep 1/N of every frozen layer per rank and time a few steps more.  The
    # headline number above is already final; should the extra variant's collectives stall, a watchdog on every rank prints
    # the line with the stall recorded in it, says so on stderr and exits NON-ZERO (3): a hang is never reported as success.
    shard_variant = None
    if world > 1 and not a.shard_frozen and not full and os.environ.get("VLB_BENCH_SHARD_VARIANT", "1") == "1":
        limit_s = float(os.environ.get("VLB_BENCH_SHARD_VARIANT_LIMIT_S", "120"))

        def bail():
            emit({"error": f"STALL: the sharded-frozen variant did not finish within {limit_s:.0f} s (collectives hung?); "
                           "headline value measured before it and unaffected; process exits with status 3"})
            print(f"bench.py[rank {rank}]: sharded-frozen variant stalled for {limit_s:.0f} s - exiting with status 3",
                  file=sys.stderr, flush=True)
            os._exit(3)
        dog = threading.Timer(limit_s, bail)
        dog.daemon = True
        dog.start()
        try:
            m.backbone.enable_sharding()
            for _ in range(2):
                step()
            barrier()
            t1 = time.perf_counter()
            ks = min(a.steps, 5)
            for _ in range(ks):
                step()
            barrier()
            d1 = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(d1, op=torch.distributed.ReduceOp.MAX)
            shard_variant = {"value": round(world * B * ks / float(d1.item()), 4), "unit": "clips/s", "steps": ks,
                             "ms_per_step": round(float(d1.item()) / ks * 1e3, 3),
                             "layout": f"frozen decoder weights sharded 1/{world} per layer, all-gathered one layer ahead (forward and reverse)"}
        except Exception as e:                                   # never lose the headline line to the extra variant
            shard_variant = {"error": f"{type(e).__name__}: {e}"[:300]}
        dog.cancel()
    emit(shard_variant)
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
