"""MX-fp8 GEMM vs the bf16 GEMM on the decoder shapes (random data, HIP events)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phantom_vlb_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, M, N, K in [("qkv", 5861, 6144, 4096), ("o", 5861, 4096, 4096), ("gate_up", 5861, 28672, 4096), ("down", 5861, 4096, 14336),
                      ("gate_up M=10240", 10240, 28672, 4096), ("sq8192", 8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
    aq, sa = ops.quantize_mxfp8(a); wq, sw = ops.quantize_mxfp8(w)
    out = torch.empty(M, N, dtype=BF, device=dev)
    t16 = t(lambda: ops.gemm(a, w, out=out)); t8 = t(lambda: ops.gemm_mxfp8(aq, sa, wq, sw, out=out)); tq = t(lambda: ops.quantize_mxfp8(a, aq, sa))
    fl = 2.0 * M * N * K
    print(f"{name:16s} bf16 {t16:.3f} ms {fl / t16 / 1e9:7.1f} TF | mxfp8 {t8:.3f} ms {fl / t8 / 1e9:7.1f} TF | quantise A [{M}x{K}] {tq * 1e3:.0f} us "
          f"({M * K * 3 / tq / 1e6:.0f} GB/s)", flush=True)
