"""2-rank data parallelism on ONE GPU (gloo backend, both ranks on cuda:0): the reduce-scattered, 1/world-scaled
gradients equal the single-process gradients on the concatenated batch, and one sharded optimiser step (each rank
updates its half of every segment, bf16 copies all-gathered) leaves both ranks with the single-process parameters.
Exercises bench.py's N>1 code path (parallel.ShardedFlatState), and `python bench.py --gpus 2` launching itself."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(lora):
    from phantom_vlb_amd.litmodule import VLBLitModuleConfig
    return VLBLitModuleConfig(model_path="none", freeze_backbone=not lora, use_lora=lora, lora_r=16 if lora else None,
                              lora_alpha=32 if lora else None, lora_dropout=0.0 if lora else None, dropout_rate=0.0,
                              num_target=128, l2_lambda=1e-3, lr=1e-3, betas=[0.9, 0.999], eps=1e-8, weight_decay=1e-2,
                              lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000, geometry="mini")


def _worker(rank, world, port, lora, ret, shard_frozen=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import vlb_oracle as O
    from phantom_vlb_amd.litmodule import VLBLitModule
    from phantom_vlb_amd.lora import LoraState
    from phantom_vlb_amd.parallel import attach_data_parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = O.geometry_mini()
        p = O.round_bf16(O.init_params(g, seed=3, lora=lora, lora_b_std=0.02))
        full = O.synthetic_batch(g, 4, seed=4)

        def build():
            m = VLBLitModule(_cfg(lora))
            m.configure_model(state_dict=p, head_state=p)
            if lora:
                m.lora = LoraState(m.geometry, m.backbone.w, 16, 32, 0.0, m.device, sd=p)
            opt, _ = m.configure_optimizers()
            return m, opt[0]
        m, opt = build()
        if shard_frozen:                  # fsdp.yaml-equivalent layout of the frozen decoder: 1/2 of every layer per rank
            full_bytes = sum(lw[k].numel() * 2 for lw in m.backbone.w.layers for k in ("wqkv", "wo", "wgu_il", "wdown"))
            m.backbone.enable_sharding()
            assert m.backbone.w.layers[0]["wqkv"] is None and m.backbone.store.world == 2
            assert m.backbone.store.shard_bytes() <= full_bytes // 2 + 64 * len(m.backbone.w.layers)
        st = attach_data_parallel(m, opt)
        assert m.world_size == 2 and st.active and st.numel * 2 == m.flat.numel and m.flat.m is None
        mine = {k: v[rank * 2:rank * 2 + 2] for k, v in full.items()}
        m.training_step(mine)
        opt.step()                        # finish reduce-scatter, clip over both ranks' slices, AdamW on the slice, all-gather
        g_dp = st.gather_full("grad")
        st.gather_masters()
        torch.cuda.synchronize()
        out = {"master": m.flat.master.cpu(), "compute": m.flat.compute.float().cpu()}
        if rank == 0:
            ref, ropt = build()
            ref.training_step(full)
            g_ref = ref.flat.grad.clone()
            ropt.step()
            torch.cuda.synchronize()
            out["err"] = float((g_dp - g_ref).abs().max() / g_ref.abs().max())
            out["perr"] = float((m.flat.master - ref.flat.master).abs().max())
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lora", [False, True])
def test_two_rank_gradients_match_single_process(dev, lora):
    import random
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, 29600 + random.randint(0, 2000), lora, ret), nprocs=2, join=True)
    assert ret[0]["err"] < (4e-2 if lora else 2e-2), ret[0]["err"]        # bf16 activations, different batch split
    assert ret[0]["perr"] < 2.5e-3, ret[0]["perr"]                         # one AdamW step of lr 1e-3: |update| <= lr
    assert torch.equal(ret[0]["master"], ret[1]["master"])                 # ranks stay in lock-step
    assert torch.equal(ret[0]["compute"], ret[1]["compute"])


def test_two_rank_lora_with_sharded_frozen_weights(dev):
    """`--shard-frozen` at world 2: every decoder layer (and its transposed copy) lives as two half shards, all-gathered
    one layer ahead in forward and in reverse for backward; gradients and the updated parameters equal the replicated run."""
    import random
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, 29600 + random.randint(0, 2000), True, ret, True), nprocs=2, join=True)
    assert ret[0]["err"] < 4e-2 and ret[0]["perr"] < 2.5e-3
    assert torch.equal(ret[0]["master"], ret[1]["master"]) and torch.equal(ret[0]["compute"], ret[1]["compute"])


def _worker_full(rank, world, port, ret, full_shard=False, fp8=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import dataclasses
    import vlb_oracle as O
    from phantom_vlb_amd.litmodule import VLBLitModule
    from phantom_vlb_amd.parallel import attach_data_parallel, sync_module_states
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = O.geometry_mini()
        p = O.round_bf16(O.init_params(g, seed=3))
        full = O.synthetic_batch(g, 4, seed=4)
        cfg = dataclasses.replace(_cfg(False), freeze_backbone=False, fp8_gemm=fp8)

        def build():
            m = VLBLitModule(cfg)
            m.configure_model(state_dict=p)
            opt, _ = m.configure_optimizers()
            return m, opt[0]
        m, opt = build()
        numel = m.full.flat.numel
        st = attach_data_parallel(m, opt, full_shard=full_shard)
        sb = m.sharded_backbone
        assert sb is not None and sb.active and sb.numel * 2 == numel and sb.grad.dtype == torch.bfloat16
        assert sb.full_shard == full_shard
        sync_module_states(m)
        mine = {k: v[rank * 2:rank * 2 + 2] for k, v in full.items()}
        if full_shard:
            # fsdp.yaml:11 FULL_SHARD: between uses this rank holds half of every layer and the tail segment only
            f = m.full.flat
            T = f.head_range[1]
            assert f.master is None and f.compute.numel() == T and f.grad.numel() == T and f.numel == numel
            assert all(m.backbone.w.layers[li][k] is None for li in range(g.layers) for k in ("wqkv", "wo", "wgu", "wdown", "in_norm"))
            assert all(lw.get(k) is None for lw in m.backbone.w.layers for k in ("wqkv_t", "wdown_t")) and not m.full.wq
            held = sb.compute.numel() + sum(b.numel() for b in sb.wpool)         # bf16 weight elements resident on this rank
            layer = max(e - s for s, e in f.layer_ranges)
            assert held == numel // 2 + 2 * layer
            assert all(p_.numel() == 0 for n, p_ in zip(opt.names, opt.param_groups[0]["params"]) if n.startswith("backbone."))
            with pytest.raises(RuntimeError, match="tail segment only"):
                f.view(f.compute, "layers.1.wdown")
        m.training_step(mine)
        opt.step()
        g_dp = sb.gather_full("grad").float()
        sb.gather_masters()
        torch.cuda.synchronize()
        if full_shard:
            out = {"master": m.full.flat.master.cpu(), "compute": sb.gather_full("compute").float().cpu(), "wt_ok": True}
            sb.release_staging()
            assert m.full.flat.master is None
            # the evaluation forward reads the layers through the same gathers (Backbone.layer_weights -> sb.get / prefetch)
            v = m.validation_step(mine)
            torch.cuda.synchronize()
            out["val_ok"] = bool(torch.isfinite(torch.as_tensor(float(v["loss"] if isinstance(v, dict) else v))))
        else:
            out = {"master": m.full.flat.master.cpu(), "compute": m.full.flat.compute.float().cpu(),
                   "wt_ok": bool(torch.equal(m.backbone.w.layers[1]["wdown_t"], m.backbone.w.layers[1]["wdown"].t()))}
        # checkpoint state under data-parallel full fine-tuning: the upstream-named state_dict must hold the UPDATED backbone
        # weights (the optimiser writes the shard buffers; trainable_state gathers them first), it must agree with the flat
        # `stores` copy, and only the writing rank builds host copies
        from phantom_vlb_amd.trainer import trainable_state
        if not full_shard:
            m.full.flat.master.zero_()                  # stale staging area: whatever survives must come from the shards
        state = trainable_state(m, 1, to_host=rank == 0)
        assert not full_shard or m.full.flat.master is None          # no standing 4 B/param copy after the checkpoint either
        if rank == 0:
            f = m.full.flat
            store = state["stores"][0]["master"]
            o, k, shp = f.offsets["layers.1.wdown"]
            out["ckpt_ok"] = bool(torch.equal(state["state_dict"]["model.layers.1.mlp.down_proj.weight"], store[o:o + k].view(shp)))
            out["ckpt_moved"] = bool((state["state_dict"]["model.layers.1.mlp.down_proj.weight"]
                                      != p["model.layers.1.mlp.down_proj.weight"]).any())
            out["ckpt_master"] = bool(torch.equal(store, out["master"]))
        else:
            out["ckpt_none"] = state is None
        if rank == 0:
            ref, ropt = build()
            ref.training_step(full)
            g_ref = ref.full.flat.grad.float().clone()
            ropt.step()
            torch.cuda.synchronize()
            out["cos"] = float((g_dp * g_ref).sum() / (g_dp.norm() * g_ref.norm()))
            out["perr"] = float((out["master"].to(ref.full.flat.master.device) - ref.full.flat.master).abs().max())
        if full_shard:
            # a second step: the refreshed slices are gathered again layer by layer, then the checkpoint is restored and the
            # step replayed - the resumed run lands on the same weights (collectives: both ranks)
            if rank == 0:
                ref.training_step(full)
                ropt.step()
            m.training_step(mine)
            opt.step()
            after2 = sb.gather_full("master").cpu()
            if rank == 0:
                torch.cuda.synchronize()
                out["perr2"] = float((after2.to(ref.full.flat.master.device) - ref.full.flat.master).abs().max())
            store = [state["stores"][0]] if rank == 0 else [None]
            dist.broadcast_object_list(store, src=0)
            for k in ("master", "m", "v"):
                opt.load_full_state(k, store[0][k], 1)
            opt.compute_from_master(1)
            out["restored"] = bool(torch.equal(sb.gather_full("master").cpu(), out["master"]))
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def test_two_rank_full_finetune_matches_single_process(dev):
    """configs[4]'s sharding in miniature: the backbone store's bf16 gradients are reduce-scattered per layer chunk,
    each rank updates its half of every segment (fp32 master / moments only there), bf16 weights are all-gathered and
    the W^T copies follow; the result tracks the single-process step on the concatenated batch."""
    import random
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_full, args=(2, 29600 + random.randint(0, 2000), ret), nprocs=2, join=True)
    assert ret[0]["cos"] > 0.995, ret[0]["cos"]                  # bf16 gradients of two half-batches vs one full batch
    assert ret[0]["perr"] < 2.5e-3
    assert torch.equal(ret[0]["master"], ret[1]["master"]) and torch.equal(ret[0]["compute"], ret[1]["compute"])
    assert ret[0]["wt_ok"] and ret[1]["wt_ok"]
    assert ret[0]["ckpt_ok"] and ret[0]["ckpt_moved"] and ret[0]["ckpt_master"] and ret[1]["ckpt_none"]


@pytest.mark.parametrize("fp8", [False, True])
def test_two_rank_full_shard_finetune_matches_single_process(dev, fp8):
    """fsdp.yaml:11 ``FULL_SHARD`` for the TRAINED weights (configs[4]'s layout in miniature): each rank keeps half of every
    decoder layer's bf16 weights, gradients, masters and moments; a layer is all-gathered into one of two buffers for its
    forward and again for its backward (W^T / MX-fp8 quantisations derived from the gathered layer), its gradients are
    reduce-scattered out of one of two layer-sized buffers.  Same numbers as the replicated-weights layout, a second step on
    re-gathered weights, evaluation forward through the gathers, checkpoint + restore."""
    import random
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_full, args=(2, 29600 + random.randint(0, 2000), ret, True, fp8), nprocs=2, join=True)
    assert ret[0]["cos"] > (0.97 if fp8 else 0.995), ret[0]["cos"]
    assert ret[0]["perr"] < 2.5e-3 and ret[0]["perr2"] < 5e-3, (ret[0]["perr"], ret[0]["perr2"])
    assert torch.equal(ret[0]["master"], ret[1]["master"]) and torch.equal(ret[0]["compute"], ret[1]["compute"])
    assert ret[0]["val_ok"] and ret[1]["val_ok"] and ret[0]["restored"] and ret[1]["restored"]
    assert ret[0]["ckpt_ok"] and ret[0]["ckpt_moved"] and ret[0]["ckpt_master"] and ret[1]["ckpt_none"]


def test_bench_gpus_2_launches_two_ranks(dev):
    """`python bench.py --gpus 2` with no launcher: two ranks are started (time-sharing this GPU over gloo), rank 0
    prints ONE JSON line with n_gpus 2; asking for more RCCL ranks than GPUs fails instead of running fewer."""
    import json
    import subprocess
    env = dict(os.environ, VLB_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--geometry", "mini", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 2 * out["config"]["clips_per_gpu"]
    assert "world=2" in out["config"]["comm"]
    sv = out["config"]["sharded_frozen_variant"]                 # the fsdp.yaml-equivalent layout is timed next to the default
    assert sv and "error" not in sv and sv["value"] > 0, sv
    env.pop("VLB_DIST_BACKEND")
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--geometry", "mini"],
                           capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
        assert r.returncode != 0 and "n_gpus" not in r.stdout


def test_bench_gpus_2_full_finetune_runs_full_shard(dev):
    """`python bench.py --gpus 2 --workload full` (mini geometry, gloo rehearsal on one GPU): the trained decoder weights run
    FULL_SHARD (fsdp.yaml:11) by default and the line says so; VLB_FSDP_STRATEGY=SHARD_GRAD_OP keeps them replicated."""
    import json
    import subprocess
    for strategy, word in ((None, "FULL_SHARD"), ("SHARD_GRAD_OP", "replicated (SHARD_GRAD_OP")):
        env = dict(os.environ, VLB_DIST_BACKEND="gloo")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VLB_FSDP_STRATEGY"):
            env.pop(k, None)
        if strategy:
            env["VLB_FSDP_STRATEGY"] = strategy
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--geometry", "mini", "--workload", "full",
                            "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert out["n_gpus"] == 2 and out["value"] > 0
        assert "trained decoder weights " + word in out["config"]["parallelism"], out["config"]["parallelism"]
