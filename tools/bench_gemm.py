"""Micro-benchmark of vlb_gemm_bf16 on the decoder / ViT / connector shapes (random data, HIP events).

  python tools/bench_gemm.py [--batch 5] [--reps 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: E402,F401  (libvlb_tools.so: variant switches / ablations live only there)
from phantom_vlb_amd import ops  # noqa: E402
from phantom_vlb_amd._lib import lib  # noqa: E402


def time_gemm(M, N, K, reps, check=False):
    dev = torch.device("cuda:0")
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, w, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(a, w, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    err = None
    if check:
        ref = (a[:512].float() @ w.float().t())
        err = float((out[:512].float() - ref).abs().max() / ref.abs().max())
    return ms, 2.0 * M * N * K / ms / 1e9, err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=5)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--variant", type=int, default=1)
    ap.add_argument("--force-tile", type=int, default=0)
    a = ap.parse_args()
    import ctypes
    lib.vlb_gemm_set_variant.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.vlb_gemm_set_variant.restype = None
    lib.vlb_gemm_set_variant(a.variant, a.force_tile)
    print(f"variant={a.variant} force_tile={a.force_tile}")
    Md = a.batch * 2048
    Mv = a.batch * 12 * 577
    shapes = [("dec qkv", Md, 6144, 4096), ("dec o", Md, 4096, 4096), ("dec gate_up", Md, 28672, 4096),
              ("dec down", Md, 4096, 14336), ("vit qkv", Mv, 3072, 1024), ("vit out", Mv, 1024, 1024),
              ("vit fc1", Mv, 4096, 1024), ("vit fc2", Mv, 1024, 4096), ("conn conv1", a.batch * 6912, 4096, 4096),
              ("conn sampler", a.batch * 1183, 4096, 32768), ("sq 4096", 4096, 4096, 4096), ("sq 8192", 8192, 8192, 8192)]
    tot_f, tot_t = 0.0, 0.0
    for name, M, N, K in shapes:
        ms, tf, err = time_gemm(M, N, K, a.reps, check=True)
        print(f"{name:14s} M={M:6d} N={N:6d} K={K:6d} kernel={lib.vlb_gemm_kernel_choice(M, N, K, 0)} "
              f"{ms:8.3f} ms  {tf:8.1f} TFLOP/s  err={err:.1e}", flush=True)
        if name.startswith("dec"):
            tot_f += 2.0 * M * N * K
            tot_t += ms
    print(f"decoder layer linears: {tot_t:.3f} ms/layer -> {tot_f / tot_t / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
