"""world_size-2 gloo tests (CPU) of the multi-GPU host logic: flat gradient bucket all-reduce,
loss scaling identity, and the sharded-layer store's gather/prefetch in both directions."""
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


def _flat_reduce(rank, world):
    from phantom_vlb_amd.parallel import FlatGradReducer
    torch.manual_seed(rank)
    a = {"w": torch.randn(7, 5), "b": torch.randn(5)}
    b = {"l": torch.randn(16, 9)}
    before = {**{k: v.clone() for k, v in a.items()}, **{k: v.clone() for k, v in b.items()}}
    red = FlatGradReducer([a, b])
    assert red.flat.numel() == 35 + 5 + 144
    assert a["w"].data_ptr() == red.flat.data_ptr()          # views, not copies
    a["b"].add_(1.0)                                         # a kernel writing a grad writes the bucket
    red()
    return {k: v.clone() for k, v in {**a, **b}.items()}, before


def test_flat_grad_reducer_sums_over_ranks():
    out = _run(_flat_reduce)
    (g0, b0), (g1, b1) = out[0], out[1]
    for k in g0:
        expect = b0[k] + b1[k] + (2.0 if k == "b" else 0.0)
        assert torch.allclose(g0[k], expect) and torch.allclose(g1[k], expect)


def _flat_reduce_overlapped(rank, world):
    """Ranges started early (as backward finishes layer groups) + the final call == one whole-bucket reduce,
    every element reduced exactly once."""
    from phantom_vlb_amd.parallel import FlatGradReducer
    torch.manual_seed(10 + rank)
    grads = {f"t{i}": torch.randn(11 + i) for i in range(9)}
    before = torch.cat([v.reshape(-1) for v in grads.values()]).clone()
    red = FlatGradReducer([grads])
    n = red.flat.numel()
    red.reduce_range(n - 30, n)             # "layers 24..31"
    red.reduce_range(40, n - 30)            # "layers 8..23"
    red.reduce_range(40, 40)                # empty range: ignored
    red()                                   # head + the remaining prefix, then waits
    assert red._pending == []
    red2 = red.flat.clone()
    red()                                   # a second call with nothing started reduces the whole bucket once more
    return red2, red.flat.clone(), before


def test_flat_grad_reducer_overlapped_ranges():
    out = _run(_flat_reduce_overlapped)
    (r0, again0, b0), (r1, again1, b1) = out[0], out[1]
    assert torch.allclose(r0, b0 + b1) and torch.allclose(r1, b0 + b1)          # each element exactly once
    assert torch.allclose(again0, 2 * (b0 + b1)) and torch.allclose(again1, again0)


def _dp_identity(rank, world):
    """sum over ranks of grad(mse_r/world + l2/world) == grad of the single-process objective on the
    concatenated batch (ridge penalty counted once) - checked with the oracle's head on CPU."""
    import vlb_oracle as O
    import torch.nn.functional as F
    from phantom_vlb_amd.parallel import FlatGradReducer, dp_loss_scales
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
    red = FlatGradReducer([mine])
    red()
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
