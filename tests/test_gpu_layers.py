"""The reference's exported head layers (src/__init__.py:3-13 - HRFConvolveLayer, RidgeRegressionLayer) as drop-in objects on
libvlb, against the oracle's restatement of src/utils.py:44-73 (itself pinned to the reference's classes, tests/test_cpu_pins.py);
the rank-ordered staged reduction of the direct reduce-scatter for world sizes 2..8 on one GPU; and the bookkeeping of the
vision side computed one step ahead (staged / skipped / consumed batches)."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", [(4, 128, 512), (3, 2048, 4096), (2, 37, 264)])
def test_hrf_convolve_layer_matches_reference_einsum(dev, dtype, shape):
    import vlb_oracle as O
    from src import HRFConvolveLayer
    B, S, E = shape
    gen = torch.Generator().manual_seed(B * 1000 + S)
    emb = torch.randn(B, S, E, generator=gen).to(dtype)
    w = torch.rand(B, S, generator=gen) * 0.2
    w[:, : S // 3] = 0                                     # the prompt / instruction span carries no weight
    ref = O.hrf_convolve(emb.float(), w)                   # fp32 math on the same (possibly bf16-valued) inputs
    out = HRFConvolveLayer()(emb.to(dev), w.to(dev))
    assert out.dtype == dtype and out.shape == (B, E)      # the reference's einsum returns the embeddings' dtype
    assert rel_err(out, ref) < (6e-3 if dtype == BF else 2e-6)   # bf16: one rounding of the output; fp32: summation order only


@pytest.mark.parametrize("B,E,V", [(5, 4096, 2048), (3, 512, 128), (11, 1024, 1000)])
def test_ridge_regression_layer_matches_reference_linear_and_penalty(dev, B, E, V):
    import vlb_oracle as O
    from src import RidgeRegressionLayer
    torch.manual_seed(B + V)
    layer = RidgeRegressionLayer(E, V, l2_lambda=1e-3, dtype=BF)
    assert layer.linear.weight.shape == (V, E) and layer.linear.weight.dtype == BF and layer.linear.bias.shape == (V,)
    x = torch.randn(B, E).to(BF)
    W, b = layer.linear.weight.float().cpu(), layer.linear.bias.float().cpu()
    ref, l2_ref = O.ridge_regression(W, b, x.float(), 1e-3)
    out, l2 = layer(x.to(dev))
    assert out.dtype == BF and out.shape == (B, V)
    assert rel_err(out, ref) < 8e-3                        # bf16 output rounding of an fp32-accumulated product
    assert abs(float(l2) - float(l2_ref)) / float(l2_ref) < 1e-5
    only = layer(x.to(dev), add_regularization=False)
    assert torch.equal(only, out)
    out32, _ = layer(x.float().to(dev))                    # fp32 activations in -> fp32 out (values already bf16-representable)
    assert out32.dtype == torch.float32 and rel_err(out32, ref) < 1e-5


@pytest.mark.parametrize("world", [2, 3, 4, 5, 6, 7, 8])
def test_staged_reduce_is_the_rank_ordered_fp32_sum(dev, world):
    """vlb_reduce_slices = the local half of vlb_reducescatter_direct(_bf16): hand-staged slices of `world` ranks summed in
    rank order - bit-equal to the same fp32 sum written with torch, for fp32 and for bf16 (fp32 accumulate, one rounding)."""
    from phantom_vlb_amd._lib import check, lib
    n = 840 * 8 * 3
    gen = torch.Generator(device=dev).manual_seed(world)
    st = torch.cuda.current_stream().cuda_stream
    stage = torch.randn(world, n, device=dev, generator=gen) * 3
    out = torch.empty(n, device=dev)
    check(lib.vlb_reduce_slices(stage.data_ptr(), out.data_ptr(), n, world, 0, st), "vlb_reduce_slices")
    ref = stage[0].clone()
    for r in range(1, world):
        ref = ref + stage[r]
    assert torch.equal(out, ref)
    stage16 = stage.to(BF)
    out16 = torch.empty(n, dtype=BF, device=dev)
    check(lib.vlb_reduce_slices(stage16.data_ptr(), out16.data_ptr(), n, world, 1, st), "vlb_reduce_slices")
    ref = stage16[0].float()
    for r in range(1, world):
        ref = ref + stage16[r].float()
    assert torch.equal(out16, ref.to(BF))
    # a bf16 ring (rounding after every hop) would differ: the single rounding is what the kernel is for
    assert lib.vlb_reduce_slices(stage16.data_ptr(), out16.data_ptr(), n + 4, world, 1, st) != 0      # not a multiple of 8: refused


def _module(dev, lora=False, full=False):
    import vlb_oracle as O
    from phantom_vlb_amd.litmodule import VLBLitModule, VLBLitModuleConfig
    cfg = VLBLitModuleConfig(model_path="none", freeze_backbone=not (lora or full), use_lora=lora, lora_r=16 if lora else None,
                             lora_alpha=32 if lora else None, lora_dropout=0.0 if lora else None, dropout_rate=0.0, num_target=128,
                             l2_lambda=1e-3, lr=1e-4, betas=[0.9, 0.999], eps=1e-8, weight_decay=1e-2,
                             lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000, geometry="mini")
    g = O.geometry_mini()
    m = VLBLitModule(cfg)
    m.configure_model(state_dict=O.round_bf16(O.init_params(g, seed=1234)))
    m.configure_optimizers()
    return m, g


def _host_batches(g, n, B=2):
    import vlb_oracle as O
    return [O.synthetic_batch(g, B, seed=100 + i) for i in range(n)]


def test_prefetched_vision_is_keyed_by_identity_not_address(dev):
    """Batches that were staged (and had their vision side started ahead) but are never consumed must not be mistaken for a
    later batch that the caching allocator places at the same address: stage -> skip -> consume gives the consumed batch's
    own result, and nothing of the skipped batches survives."""
    from phantom_vlb_amd.datamodule import DevicePrefetcher
    m, g = _module(dev)
    bb = m.backbone
    batches = _host_batches(g, 6)
    want = [float(m.validation_step({k: (v.to(dev) if torch.is_tensor(v) and k not in ("language", "padvals") else v)
                                     for k, v in b.items()})["loss"]) for b in batches]
    assert len(set(want)) == len(want)                     # the batches do differ
    loader = DevicePrefetcher(batches, dev, on_staged=m.prefetch_vision, on_discard=m.discard_prefetched_vision)
    # (1) rank-strided selection: only every second batch is staged at all; each consumed batch gets ITS result
    got = {}
    for bi, batch in loader.iter_selected(lambda i: i % 2 == 1):
        got[bi] = float(m.validation_step(batch)["loss"])
        del batch                                          # free the pixels: the next staged batch may reuse the address
    assert got == {1: want[1], 3: want[3], 5: want[5]}
    assert not bb._vis_queue and not bb._vis_pending
    # (2) the consumer stops early (limit_val_batches / max_steps): the batch staged ahead is discarded, not left behind
    it = loader.iter_selected()
    bi, batch = next(it)
    l0 = float(m.validation_step(batch)["loss"])           # launches batch 1's vision side behind this forward
    assert l0 == want[0] and len(bb._vis_queue) + len(bb._vis_pending) == 1
    it.close()
    assert not bb._vis_queue and not bb._vis_pending
    # (3) the old failure: a result computed for pixels that were freed, then a NEW tensor at the same address
    stale = batches[2]["vision"].to(dev)
    bb.prefetch_video_tokens(stale)
    addr = stale.data_ptr()
    bb.discard_video_tokens(stale)                          # the batch is skipped
    del stale
    fresh = batches[4]["vision"].to(dev)
    b4 = dict(batches[4], vision=fresh, timeseries=batches[4]["timeseries"].to(dev))
    assert float(m.validation_step(b4)["loss"]) == want[4]
    if fresh.data_ptr() == addr:                           # the allocator did hand the block back: the case the advisor described
        bb.prefetch_video_tokens(fresh)                    # without a discard an identity-keyed entry could still never match another tensor
        other = batches[5]["vision"].to(dev)
        assert bb.video_tokens(other) is not None and len(bb._vis_queue) == 1
        bb.discard_video_tokens()
    torch.cuda.synchronize()


def test_full_finetune_validation_reuses_the_prefetched_tower(dev):
    """Full fine-tune: the tower alone is computed ahead (the connector trains).  The EVAL forward wants the finished video
    tokens: it takes the prefetched tower features and runs the connector in line - the tower is not run twice and no
    queue entry is left behind."""
    m, g = _module(dev, full=True)
    bb = m.backbone
    b = _host_batches(g, 1)[0]
    batch = {k: (v.to(dev) if torch.is_tensor(v) and k not in ("language", "padvals") else v) for k, v in b.items()}
    want = float(m.validation_step(batch)["loss"])
    calls = []
    tower = bb.vision_tower
    bb.vision_tower = lambda x: (calls.append(1), tower(x))[1]
    m.prefetch_vision(batch)
    bb.launch_deferred_video_tokens()
    assert len(bb._vis_queue) == 1 and len(calls) == 1
    got = float(m.validation_step(batch)["loss"])
    assert got == want and len(calls) == 1 and not bb._vis_queue and not bb._vis_pending
    m.prefetch_vision(batch)                               # and the training step still finds its tower features
    bb.launch_deferred_video_tokens()
    m.training_step(batch)
    assert len(calls) == 2 and not bb._vis_queue
    torch.cuda.synchronize()
