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
