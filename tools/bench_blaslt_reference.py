"""Measurement-only yardstick: what the vendor library (torch.matmul -> hipBLASLt) reaches on the
decoder GEMM shapes, next to libvlb's kernel.  Not used by the product path."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phantom_vlb_amd import ops

dev = torch.device("cuda")
shapes = [(10240, 28672, 4096, "gate/up B=5"), (10240, 6144, 4096, "qkv"), (10240, 4096, 4096, "o"),
          (10240, 4096, 14336, "down"), (6144, 28672, 4096, "gate/up B=3"), (6144, 4096, 14336, "down B=3"),
          (8192, 8192, 8192, "8192^3")]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


if os.environ.get("VLB_SHAPES") == "lora":     # the LoRA batch (5861 packed rows): forward and dgrad shapes of the decoder
    shapes = [(5861, 6144, 4096, "qkv"), (5861, 4096, 4096, "o"), (5861, 28672, 4096, "gate/up"), (5861, 4096, 14336, "down"),
              (5861, 4096, 6144, "dgrad qkv"), (5861, 4096, 28672, "dgrad gate/up"), (5861, 14336, 4096, "dgrad down")]
for M, N, K, name in shapes:
    a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    t_lt = timeit(lambda: torch.matmul(a, w.t()))
    t_vlb = timeit(lambda: ops.gemm(a, w))
    fl = 2.0 * M * N * K / 1e9
    print(f"{name:14s} M={M} N={N} K={K}: hipBLASLt {t_lt:.3f} ms = {fl / t_lt:7.1f} TF | libvlb {t_vlb:.3f} ms = {fl / t_vlb:7.1f} TF", flush=True)
