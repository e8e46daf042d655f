"""world_size-2 gloo tests (CPU) of the multi-GPU host logic: reduce-scatter / slice update / all-gather of the
sharded flat optimiser state, the loss scaling identity, and the sharded-layer store's gather/prefetch in both
directions."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    import random
    port = 29500 + random.randint(0, 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, fn, ret), nprocs=world, join=True)
    return dict(ret)


class _FakeFlat:
    """CPU stand-in for flat.FlatTrainables: the five flat buffers + the segment table ShardedFlatState reads."""

    def __init__(self, layers=4, head=6720, per_layer=6720 * 2, seed=0):
        n = head + layers * per_layer
        gen = torch.Generator().manual_seed(seed)
        self.numel = n
        self.master = torch.randn(n, generator=gen)
        self.compute = self.master.bfloat16()
        self.grad = torch.zeros(n)
        self.m, self.v = torch.zeros(n), torch.zeros(n)
        self.head_range = (0, head)
        self.layer_ranges = [(head + i * per_layer, head + (i + 1) * per_layer) for i in range(layers)]


def _sharded_state(rank, world):
    """reduce-scatter under 'backward' (two layer chunks started early) + the rest at finish: every rank ends up with
    the summed gradient of exactly its slices; a slice-local update + all-gather rebuilds identical full bf16 copies;
    masters round-trip through gather / load."""
    from phantom_vlb_amd.parallel import ShardedFlatState
    flat = _FakeFlat()
    st = ShardedFlatState(flat, chunks=2)
    assert st.world == world and st.numel * world == flat.numel and flat.m is None
    assert len(st.segments) == 3 and st.layer_seg == {0: 1, 2: 2}
    torch.manual_seed(100 + rank)
    flat.grad.copy_(torch.randn(flat.numel))
    local = flat.grad.clone()
    for li in (3, 2):                 # backward walks down: chunk [2,3] is final when layer 2 is done
        st.on_layer_done(li)
    assert list(st._pending) == [2]
    st.on_layer_done(1)
    st.on_layer_done(0)
    st.finish_reduce()
    summed = st.gather_full("grad")
    # "optimiser": master_slice -= 0.5 * grad_slice, bf16 copy refreshed, then gathered
    st.master.sub_(0.5 * st.grad)
    st.compute.copy_(st.master)
    st.gather_compute()
    st.gather_masters()
    m_after = flat.master.clone()
    # masters: scramble the staging copy's foreign slices, reload own slices, gather again -> unchanged
    st.load_masters()
    st.gather_masters()
    return local, summed, m_after, flat.compute.float().clone(), flat.master.clone()


def test_sharded_flat_state_reduce_scatter_update_all_gather():
    out = _run(_sharded_state)
    ref = _FakeFlat()
    total = out[0][0] + out[1][0]
    for r in (0, 1):
        local, summed, m_after, comp, m_again = out[r]
        assert torch.allclose(summed, total)
        assert torch.allclose(m_after, ref.master - 0.5 * total, atol=1e-6)
        assert torch.equal(comp, m_after.bfloat16().float())
        assert torch.equal(m_again, m_after)
    assert torch.equal(out[0][2], out[1][2])


class _FakeFullShardFlat(_FakeFlat):
    """+ the two methods parallel.ShardedFlatState._enter_full_shard asks of fullft.FlatBackbone."""

    def __init__(self, **kw):
        super().__init__(**kw)
        self.grad = torch.zeros(self.numel, dtype=torch.bfloat16)

    def release_layers(self):
        T = self.head_range[1]
        self.compute, self.grad, self.master = self.compute[:T].clone(), torch.zeros(T, dtype=torch.bfloat16), None

    def layer_views(self, buf, li):
        return {"w": buf}


def _full_shard_state(rank, world):
    """fsdp.yaml:11 FULL_SHARD for trained weights, the host logic on CPU tensors: after the state is attached a rank holds its
    half of every layer + the tail; layer_weights() gathers a layer into one of two buffers (the next one started ahead, in both
    walking directions), the backward's gradients go through two rotating layer buffers into the owned slices, the update's
    refreshed slices come back through the per-layer gathers."""
    from phantom_vlb_amd.parallel import ShardedFlatState
    flat = _FakeFullShardFlat()
    L, (T0, T1) = len(flat.layer_ranges), flat.head_range
    full_w = flat.compute.clone()
    st = ShardedFlatState(flat, full_shard=True)
    assert st.full_shard and len(st.segments) == L + 1 and st.layer_seg == {i: i + 1 for i in range(L)}
    assert flat.master is None and flat.compute.numel() == T1 and flat.grad.numel() == T1
    assert st.compute.numel() * world == full_w.numel() and len(st.wpool) == 2 and len(st.gpool) == 2
    ok = True
    for li in list(range(L)) + list(range(L - 1, -1, -1)):                    # forward order, then backward order
        nxt = li + 1 if li + 1 < L else None
        s, e = flat.layer_ranges[li]
        ok &= bool(torch.equal(st.layer_weights(li, then=nxt), full_w[s:e]))
    ok &= bool(torch.equal(st.get(1)["w"], full_w[slice(*flat.layer_ranges[1])]))
    # backward: every rank writes its own gradient of layer li into the buffer it is handed, then reports the layer done
    gens = [torch.Generator().manual_seed(500 + r) for r in range(world)]
    grads = [torch.randn(full_w.numel(), generator=g_).bfloat16() for g_ in gens]          # what each rank "computes"
    for li in range(L - 1, -1, -1):
        s, e = flat.layer_ranges[li]
        st.layer_grads(li).copy_(grads[rank][s:e])
        st.on_layer_done(li)
    flat.grad.copy_(grads[rank][T0:T1])
    st.finish_reduce()
    total = sum(g_.float() for g_ in grads)
    for si in range(L + 1):
        whole, mine = st._own(si)
        ok &= bool(torch.allclose(st.grad[mine].float(), total[whole], atol=0.05, rtol=0.02))
    # "optimiser": a slice-local update, then the tail is gathered and the layers come back layer by layer
    st.master.sub_(0.5 * st.grad.float())
    st.compute.copy_(st.master)
    st.gather_compute()
    want = st.gather_full("compute")
    ok &= bool(torch.equal(flat.compute, want[T0:T1]))
    for li in range(L):
        s, e = flat.layer_ranges[li]
        ok &= bool(torch.equal(st.layer_weights(li, then=li + 1), want[s:e]))
    ok &= bool((want[flat.layer_ranges[0][0]:].float() != full_w[flat.layer_ranges[0][0]:].float()).any())
    # staging copy for a checkpoint, dropped again; masters restored from a full-size copy
    st.gather_masters()
    staged = flat.master.clone()
    st.release_staging()
    ok &= flat.master is None and staged.numel() == full_w.numel()
    st.master.zero_()
    st.load_full("master", staged)
    st.compute_from_master()
    ok &= bool(torch.equal(st.gather_full("compute"), want))
    return bool(ok)


def test_full_shard_state_gathers_layers_and_reduces_through_rotating_buffers():
    ret = _run(_full_shard_state)
    assert ret[0] and ret[1]


def test_sharded_flat_state_world_one_aliases_the_flat_buffers():
    from phantom_vlb_amd.parallel import ShardedFlatState
    flat = _FakeFlat()
    st = ShardedFlatState(flat)
    assert st.world == 1 and not st.active
    assert st.master.data_ptr() == flat.master.data_ptr() and st.grad.data_ptr() == flat.grad.data_ptr()
    st.on_layer_done(0); st.finish_reduce(); st.gather_compute()        # all no-ops
    assert st.gather_full("m") is flat.m


def test_sharded_flat_state_rejects_world_sizes_that_do_not_divide_the_segments():
    import pytest
    from phantom_vlb_amd.parallel import ShardedFlatState

    class _Comm:
        world, rank = 11, 0
    with pytest.raises(ValueError):
        ShardedFlatState(_FakeFlat(), comm=_Comm())


def _dp_identity(rank, world):
    """sum over ranks of grad(mse_r/world + l2/world) == grad of the single-process objective on the
    concatenated batch (ridge penalty counted once) - checked with the oracle's head on CPU."""
    import vlb_oracle as O
    import torch.nn.functional as F
    from phantom_vlb_amd.parallel import dp_loss_scales
    g = O.Geometry(dim=32, num_target=16, l2_lambda=1e-2)
    gen = torch.Generator().manual_seed(0)
    p = {"layer_norm1.weight": torch.ones(32), "layer_norm1.bias": torch.zeros(32), "layer_norm2.weight": torch.ones(32),
         "layer_norm2.bias": torch.zeros(32), "ridge_layer.linear.weight": torch.randn(16, 32, generator=gen) * 0.2,
         "ridge_layer.linear.bias": torch.zeros(16)}
    hidden, wm, y = torch.randn(4, 10, 32, generator=gen), torch.rand(4, 10, generator=gen), torch.randn(4, 16, generator=gen)

    def grads(h, w, t, ms, ls):
        q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        pred, l2, _ = O.brain_head(q, h, w, g)
        (ms * F.mse_loss(pred, t) + ls * l2).backward()
        return {k: v.grad for k, v in q.items()}
    full = grads(hidden, wm, y, 1.0, 1.0)
    ms, ls = dp_loss_scales(world)
    sl = slice(rank * 2, rank * 2 + 2)
    mine = grads(hidden[sl], wm[sl], y[sl], ms, ls)
    for k in mine:
        dist.all_reduce(mine[k])
    return max(float((mine[k] - full[k]).abs().max()) for k in full)


def test_data_parallel_gradient_identity():
    out = _run(_dp_identity)
    assert out[0] < 1e-6 and out[1] < 1e-6


def _store(rank, world):
    from phantom_vlb_amd.parallel import ShardedLayerStore
    torch.manual_seed(0)
    layers = [{"wqkv": torch.randn(12, 8).bfloat16(), "wo": torch.randn(8, 8).bfloat16(),
               "wgu": torch.randn(20, 8).bfloat16(), "wdown": torch.randn(8, 10).bfloat16(), "norm": torch.ones(8)}
              for _ in range(5)]
    store = ShardedLayerStore(layers, ("wqkv", "wo", "wgu", "wdown"))
    full_bytes = sum(sum(l[k].numel() for k in ("wqkv", "wo", "wgu", "wdown")) for l in layers) * 2
    ok = store.shard_bytes() <= full_bytes // world + 5 * 64
    for order in (range(5), range(4, -1, -1)):            # forward order, then backward order
        step = 1 if order[0] == 0 else -1
        for i in order:
            w = store.get(i)
            store.prefetch(i + step)
            for k in ("wqkv", "wo", "wgu", "wdown"):
                ok = ok and torch.equal(w[k], layers[i][k])
    return bool(ok)


def test_sharded_layer_store_roundtrip():
    out = _run(_store)
    assert out[0] and out[1]


def _validate_with_an_idle_rank(rank, world):
    """Rank-strided validation with fewer batches than ranks (limit_val_batches=1 at world 2): rank 1 draws nothing and
    must still issue the callback's collectives - the Pearson sums and the loss agree on both ranks afterwards."""
    from phantom_vlb_amd.trainer import Trainer
    from phantom_vlb_amd.utils import LogValAccuracyCallback

    class _Cfg:
        num_target = 5

    class _Model:
        config, device, backbone = _Cfg(), torch.device("cpu"), object()

        def __init__(self):
            self.logged = {}

        def log(self, k, v, **kw):
            self.logged[k] = v

        def validation_step(self, batch):
            return {"loss": batch["y"].pow(2).mean(), "brain_preds": batch["p"], "brain_vals": batch["y"]}

    gen = torch.Generator().manual_seed(7)
    batches = [{"p": torch.randn(6, 5, generator=gen), "y": torch.randn(6, 5, generator=gen)} for _ in range(3)]
    cb = LogValAccuracyCallback()
    tr = Trainer(callbacks=[cb], limit_val_batches=1, devices=world)
    m = _Model()
    metrics = tr.validate(m, batches)
    ref = torch.stack([torch.corrcoef(torch.stack([batches[0]["p"][:, i], batches[0]["y"][:, i]]))[0, 1] for i in range(5)])
    return {"loss": metrics["val/brain_loss"], "corr": cb.correlations.clone(), "n": cb.n,
            "err": float((cb.correlations - ref).abs().max()), "want": float(batches[0]["y"].pow(2).mean())}


def test_validation_with_fewer_batches_than_ranks_keeps_collectives_aligned():
    r = _run(_validate_with_an_idle_rank)
    assert r[0]["n"] == r[1]["n"] == 6 and torch.equal(r[0]["corr"], r[1]["corr"])
    assert r[0]["err"] < 1e-5 and abs(r[0]["loss"] - r[0]["want"]) < 1e-6 and r[0]["loss"] == r[1]["loss"]
