"""Achieved HBM GB/s of the memory-bound kernels at the 7B / B=5 shapes (algorithmic bytes / HIP-event time).

  python tools/bench_hbm_kernels.py            # prints a table; peak HBM3E = 8000 GB/s spec, ~6300 achievable
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phantom_vlb_amd import ops  # noqa: E402
from phantom_vlb_amd._lib import check, lib  # noqa: E402
from phantom_vlb_amd.head import BrainHead  # noqa: E402

BF = torch.bfloat16
dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def row(name, nbytes, ms):
    print(f"{name:34s} {nbytes / 1e6:10.1f} MB  {ms * 1e3:9.1f} us  {nbytes / ms / 1e6:8.0f} GB/s  {nbytes / ms / 1e6 / 8000 * 100:5.1f}% of 8 TB/s", flush=True)


def main():
    B, S, E, V, FF = 5, 2048, 4096, 2048, 14336
    M = B * S
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(M, E, device=dev, generator=g).to(BF)
    w = torch.ones(E, device=dev, dtype=BF)
    y = torch.empty_like(x)
    row("rmsnorm_fwd [10240x4096]", 2 * x.numel() * 2, timeit(lambda: ops.rmsnorm(x, w, 1e-5, out=y)))
    row("rmsnorm_bwd", 3 * x.numel() * 2, timeit(lambda: ops.rmsnorm_bwd(x, w, x, 1e-5, out=y)))
    xv = torch.randn(B * 12 * 577, 1024, device=dev, generator=g).to(BF)
    wv = torch.ones(1024, device=dev, dtype=BF)
    yv = torch.empty_like(xv)
    row("layernorm_fwd ViT [34620x1024]", 2 * xv.numel() * 2, timeit(lambda: ops.layernorm(xv, wv, wv, 1e-5, out=yv)))
    xc = torch.randn(B * 12 * 576, 4096, device=dev, generator=g).to(BF)
    yc = torch.empty_like(xc)
    row("layernorm+res+silu conn [34560x4096]", 3 * xc.numel() * 2, timeit(lambda: ops.layernorm(xc, w, w, 1e-6, residual=xc, act=3, out=yc)))
    qkv = torch.randn(M, 6144, device=dev, generator=g).to(BF)
    cos = torch.rand(S, 64, device=dev)
    row("rope q|k in place [10240x5120]", 2 * M * 5120 * 2, timeit(lambda: ops.rope_(qkv, cos, cos, B, S, 40, 128)))
    gu = torch.randn(M, 2 * FF, device=dev, generator=g).to(BF)
    hh = torch.empty(M, FF, device=dev, dtype=BF)
    row("swiglu_fwd [10240x28672]", 3 * M * FF * 2, timeit(lambda: ops.swiglu(gu, out=hh)))
    dgu = torch.empty_like(gu)
    row("swiglu_bwd", 5 * M * FF * 2, timeit(lambda: ops.swiglu_bwd(gu, hh, out=dgu)))
    vis = torch.randn(B * 12, 3, 336, 336, device=dev, generator=g)
    row("patchify fp32->bf16 (+pad 588->640)", vis.numel() * 4 + B * 12 * 576 * 640 * 2, timeit(lambda: ops.patchify(vis, 14, 640)))
    row("dwconv3x3 [60x24x24x4096]", 2 * xc.numel() * 2, timeit(lambda: ops.dwconv3x3(xc, torch.ones(9, 4096, device=dev, dtype=BF), B * 12, 24, 24, 4096)))
    row("se_pool", xc.numel() * 2, timeit(lambda: ops.se_pool(xc, B * 12, 576, 4096)))
    row("im2col3d (k2 s2 p1)", xc.numel() * 2 + B * 1183 * 8 * 4096 * 2, timeit(lambda: ops.im2col3d(xc, B, 12, 24, 24, 4096)))
    ids = torch.randint(3, 32000, (B, 866), device=dev, generator=g)
    ids[:, 100] = -201
    emb = torch.randn(32000, E, device=dev, generator=g).to(BF)
    vid = torch.randn(B * 1183, E, device=dev, generator=g).to(BF)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    row("splice_embed", 2 * M * E * 2, timeit(lambda: ops.splice_embed(ids, emb, vid, 1183, -201, err)))
    # ---- head
    for Vh in (2048, 65536):
        head = BrainHead(E, Vh, 1e-3, 1e-5, dev)
        wm = torch.rand(B, S, device=dev, generator=g)
        wm[:, :700] = 0                                    # prompt/instruction span: skipped rows are not read
        yy = torch.randn(B, Vh, device=dev, generator=g)
        read_rows = int((wm != 0).sum())
        t_f = timeit(lambda: head.forward(x, wm, yy), reps=5)
        row(f"head_fwd V={Vh} (pool+LN2+ridge+loss)", read_rows * E * 2 + Vh * E * 2, t_f)
        t_b = timeit(lambda: head.backward(need_dhidden=False), reps=5)
        row(f"head_bwd V={Vh} (dW fp32 + dz + LN grads)", 2 * Vh * E * 2 + Vh * E * 4, t_b)
        t_h = timeit(lambda: head.backward(need_dhidden=True), reps=5) - t_b
        row(f"  + d hidden (LoRA only)", read_rows * E * 2 + M * E * 2, t_h)
        n = Vh * E
        mst, grd, m1, v1 = (torch.zeros(n, device=dev) for _ in range(4))
        cp = torch.zeros(n, device=dev, dtype=BF)
        ss = torch.zeros(1, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        row(f"adamw_step V={Vh} ({n / 1e6:.0f} M params)", n * 30,
            timeit(lambda: check(lib.vlb_adamw_step(mst.data_ptr(), cp.data_ptr(), grd.data_ptr(), m1.data_ptr(), v1.data_ptr(), n,
                                                    1e-4, 0.9, 0.999, 1e-8, 1e-2, 1, ss.data_ptr(), 1.0, st), "adamw"), reps=5))
        del head, mst, grd, m1, v1, cp
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
