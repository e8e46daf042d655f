"""2-rank data parallelism on ONE GPU (gloo backend, both ranks on cuda:0): the summed, 1/world-scaled
gradients equal the single-process gradients on the concatenated batch, and one optimiser step leaves
both ranks with identical parameters.  Exercises bench.py's N>1 code path (FlatTrainables bucket)."""
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


def _worker(rank, world, port, lora, ret):
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
        attach_data_parallel(m, opt)
        assert m.world_size == 2
        mine = {k: v[rank * 2:rank * 2 + 2] for k, v in full.items()}
        m.training_step(mine)
        opt.grad_reducer()
        g_dp = m.flat.grad.clone()
        opt.grad_reducer = None           # already reduced above
        opt.step()
        torch.cuda.synchronize()
        out = {"master": m.flat.master.cpu()}
        if rank == 0:
            ref, _ = build()
            ref.training_step(full)
            g_ref = ref.flat.grad
            out["err"] = float((g_dp - g_ref).abs().max() / g_ref.abs().max())
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
    assert torch.equal(ret[0]["master"], ret[1]["master"])                 # ranks stay in lock-step
