"""vlb_comm_* (libvlb's own RCCL entry points) on the one GPU of the test box: communicator of world size 1 -
unique id, init, the direct all-gather / reduce-scatter / scalar all-reduce, and a whole sharded optimiser step
driven through them (ShardedFlatState with force_collectives) equal to the plain single-process step.
More than one RCCL rank needs one GPU per rank: the N-rank behaviour is covered by the gloo tests
(test_cpu_parallel.py, test_gpu_data_parallel.py) through the same ShardedFlatState code."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_direct_collectives_world_one(dev):
    from phantom_vlb_amd.parallel_native import DirectComm
    c = DirectComm()
    assert (c.world, c.rank) == (1, 0)
    x = torch.randn(4096, device=dev)
    out = torch.zeros(4096, device=dev)
    c.reduce_scatter(out, x).wait()
    full = torch.zeros(1000, dtype=torch.bfloat16, device=dev)
    shard = torch.randn(1000, device=dev).bfloat16()
    c.all_gather(full, shard).wait()
    s = torch.tensor([3.5], device=dev)
    c.all_reduce_scalar(s)
    torch.cuda.synchronize()
    assert torch.equal(out, x) and torch.equal(full, shard) and float(s) == 3.5
    # in-place gather (the shard already sits at its slot of the full buffer)
    c.all_gather(full, full).wait()
    torch.cuda.synchronize()
    assert torch.equal(full, shard)


def test_sharded_step_through_direct_comm_equals_plain_step(dev):
    import vlb_oracle as O
    from phantom_vlb_amd.litmodule import VLBLitModule, VLBLitModuleConfig
    from phantom_vlb_amd.parallel import attach_data_parallel
    from phantom_vlb_amd.parallel_native import DirectComm
    cfg = VLBLitModuleConfig(model_path="none", freeze_backbone=False, use_lora=True, lora_r=16, lora_alpha=32, lora_dropout=0.1,
                             dropout_rate=0.1, num_target=128, l2_lambda=1e-3, lr=1e-3, betas=[0.9, 0.999], eps=1e-8,
                             weight_decay=1e-2, lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000, geometry="mini")
    g = O.geometry_mini()
    p = O.round_bf16(O.init_params(g, seed=3, lora=True, lora_b_std=0.02))
    batch = O.synthetic_batch(g, 4, seed=4)
    outs = []
    for direct in (False, True):
        m = VLBLitModule(cfg)
        m.configure_model(state_dict=p)
        opt, _ = m.configure_optimizers()
        if direct:
            st = attach_data_parallel(m, opt[0], comm=DirectComm(), force_collectives=True)
            assert st.active and st.master.data_ptr() != m.flat.master.data_ptr()
        for _ in range(2):
            m.training_step(batch)
            opt[0].step()
        if direct:
            st.gather_masters()
        torch.cuda.synchronize()
        outs.append((m.flat.master.clone(), m.flat.compute.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("fp8", [False, True])
def test_full_shard_step_through_direct_comm_equals_plain_step(dev, fp8):
    """fsdp.yaml:11 FULL_SHARD of the TRAINED decoder weights with the RCCL calls in the loop (libvlb's communicator at
    world 1, ``force_collectives``): per-layer all-gathers into the two rotating buffers, W^T (bf16) / MX quantisations (fp8)
    derived per gathered layer, gradients reduce-scattered out of the rotating gradient buffers.  At world 1 every
    collective is an identity, so three optimiser steps must leave BIT-identical masters, moments and bf16 weights to the
    plain single-process run that keeps everything resident."""
    import dataclasses
    import vlb_oracle as O
    from phantom_vlb_amd.litmodule import VLBLitModule, VLBLitModuleConfig
    from phantom_vlb_amd.parallel import attach_data_parallel
    from phantom_vlb_amd.parallel_native import DirectComm
    cfg = VLBLitModuleConfig(model_path="none", freeze_backbone=False, use_lora=False, lora_r=None, lora_alpha=None, lora_dropout=None,
                             dropout_rate=0.1, num_target=128, l2_lambda=1e-3, lr=1e-3, betas=[0.9, 0.999], eps=1e-8,
                             weight_decay=1e-2, lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000, geometry="mini",
                             fp8_gemm=fp8)
    g = O.geometry_mini()
    p = O.round_bf16(O.init_params(g, seed=3))
    batch = O.synthetic_batch(g, 4, seed=4)
    outs = []
    for shard in (False, True):
        m = VLBLitModule(cfg)
        m.configure_model(state_dict=p)
        opt, _ = m.configure_optimizers()
        sb = None
        if shard:
            attach_data_parallel(m, opt[0], comm=DirectComm(), force_collectives=True, full_shard=True)
            sb = m.sharded_backbone
            assert sb.full_shard and m.full.flat.master is None and m.backbone.w.layers[0]["wqkv"] is None
        losses = []
        for _ in range(3):
            losses.append(float(m.training_step(batch)))
            opt[0].step()
        val = m.validation_step(batch)
        torch.cuda.synchronize()
        if shard:
            outs.append((sb.gather_full("master"), sb.gather_full("compute"), sb.gather_full("m"), sb.gather_full("v"), losses, val))
        else:
            f = m.full.flat
            outs.append((f.master.clone(), f.compute.clone(), f.m.clone(), f.v.clone(), losses, val))
    for a, b in zip(outs[0][:4], outs[1][:4]):
        assert torch.equal(a, b)
    assert outs[0][4] == outs[1][4]
    va, vb = outs[0][5], outs[1][5]
    assert float(va["loss"] if isinstance(va, dict) else va) == float(vb["loss"] if isinstance(vb, dict) else vb)


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_direct_schedules_on_the_loopback_transport(dev, world):
    """The all-pairs exchange code of vlb_allgather_direct / vlb_reducescatter_direct(_bf16) / vlb_allreduce_scalar - peer
    loops, slice offsets, staging layout, rank-ordered reduction - with world 2..8 on ONE GPU: `world` communicators in this
    process (vlb_comm_init_loopback: ncclSend / ncclRecv replaced by stream-ordered device copies paired through a mailbox),
    every rank driven by its own host thread through the same DirectComm wrapper the RCCL path uses.  Every rank's result is
    compared with a torch reference built from ALL ranks' inputs: reduce-scatter bit-exact against the rank-ordered fp32 sum
    (bf16 slices: fp32 accumulation, one rounding), all-gather exact (also in place), scalar all-reduce exact.  Two rounds
    per collective, so FIFO pairing and the all-reduce's double-buffered staging are exercised too.  (RCCL itself has only
    ever run at world 1 here: no multi-GPU box.)"""
    import threading
    from phantom_vlb_amd.parallel_native import DirectComm
    comms = DirectComm.loopback(world)
    assert [c.rank for c in comms] == list(range(world)) and all(c.world == world for c in comms)
    n, nb, m = 4 * 1531, 8 * 517, 1000          # per-rank slice sizes (fp32 reduce-scatter, bf16 reduce-scatter, all-gather shard)
    gen = torch.Generator(device=dev).manual_seed(100 + world)
    rounds = 2
    x = [[torch.randn(world * n, device=dev, generator=gen) for _ in range(world)] for _ in range(rounds)]
    y = [[torch.randn(world * nb, device=dev, generator=gen).bfloat16() for _ in range(world)] for _ in range(rounds)]
    sh = [[torch.randn(m, device=dev, generator=gen).bfloat16() for _ in range(world)] for _ in range(rounds)]
    sc = [[torch.randn(3, device=dev, generator=gen) for _ in range(world)] for _ in range(rounds)]
    res = [[None] * world for _ in range(rounds)]
    errors = []

    def run(r):
        try:
            torch.cuda.set_device(dev)
            c = comms[r]
            for k in range(rounds):
                out = torch.zeros(n, device=dev)
                outb = torch.zeros(nb, dtype=torch.bfloat16, device=dev)
                full = torch.zeros(world * m, dtype=torch.bfloat16, device=dev)
                inplace = torch.zeros(world * m, dtype=torch.bfloat16, device=dev)
                inplace[r * m:(r + 1) * m] = sh[k][r]
                s = sc[k][r].clone()
                c.reduce_scatter(out, x[k][r]).wait()
                c.reduce_scatter(outb, y[k][r]).wait()
                c.all_gather(full, sh[k][r]).wait()
                c.all_gather(inplace, inplace[r * m:(r + 1) * m]).wait()
                c.all_reduce_scalar(s)
                res[k][r] = (out, outb, full, inplace, s)
            torch.cuda.synchronize()
        except Exception as e:       # surfaced below: a failing rank must not leave the others waiting silently
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=180)
    assert not errors and not any(t.is_alive() for t in threads), errors
    torch.cuda.synchronize()
    for k in range(rounds):
        gathered = torch.cat(sh[k])
        ssum = sc[k][0].clone()
        for q in range(1, world):
            ssum = ssum + sc[k][q]
        for r in range(world):
            out, outb, full, inplace, s = res[k][r]
            ref = x[k][0][r * n:(r + 1) * n].clone()
            refb = y[k][0][r * nb:(r + 1) * nb].float()
            for q in range(1, world):                       # rank order, fp32
                ref = ref + x[k][q][r * n:(r + 1) * n]
                refb = refb + y[k][q][r * nb:(r + 1) * nb].float()
            assert torch.equal(out, ref), (world, k, r, "reduce-scatter fp32")
            assert torch.equal(outb, refb.bfloat16()), (world, k, r, "reduce-scatter bf16")
            assert torch.equal(full, gathered) and torch.equal(inplace, gathered), (world, k, r, "all-gather")
            assert torch.equal(s, ssum), (world, k, r, "all-reduce")
    del comms
