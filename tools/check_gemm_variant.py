"""Correctness sweep of one GEMM kernel variant against torch fp32 (run before benchmarking it)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: E402,F401  (libvlb_tools.so: variant switches / ablations live only there)
from phantom_vlb_amd import ops  # noqa: E402
from phantom_vlb_amd._lib import lib  # noqa: E402

lib.vlb_gemm_set_variant.argtypes = [ctypes.c_int, ctypes.c_int]
lib.vlb_gemm_set_variant.restype = None
v = int(sys.argv[1])
dev = torch.device("cuda:0")
ok = True
for rep in range(3):
    for (M, N, K, K2) in [(512, 256, 128, 0), (1000, 512, 192, 0), (300, 256, 64, 0), (2048, 1024, 640, 0),
                          (768, 512, 256, 64), (4096, 4096, 4096, 0), (10240, 6144, 4096, 0), (1184, 256, 32, 0),
                          (5000, 768, 96, 32), (256, 256, 32, 0), (256, 512, 64, 0), (260, 256, 160, 0),
                          (4500, 4096, 256, 0), (4500, 4096, 4096, 64), (5861, 4096, 4096, 0)]:   # tail-split cuts
        lib.vlb_gemm_set_variant(v, 0)
        g = torch.Generator(device=dev).manual_seed(M + N + K + rep)
        a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev, generator=g) * 0.1).to(torch.bfloat16)
        a2 = w2 = None
        ref = a.float() @ w.float().t()
        if K2:
            a2 = torch.randn(M, K2, device=dev, generator=g).to(torch.bfloat16)
            w2 = (torch.randn(N, K2, device=dev, generator=g) * 0.1).to(torch.bfloat16)
            ref = ref + a2.float() @ w2.float().t()
        bias = torch.randn(N, device=dev, generator=g).to(torch.bfloat16)
        res = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
        out = ops.gemm(a, w, bias=bias, residual=res, a2=a2, w2=w2)
        ref = ref + bias.float() + res.float()
        err = float((out.float() - ref).abs().max() / ref.abs().max())
        choice = lib.vlb_gemm_kernel_choice(M, N, K, K2)
        flag = "OK " if err < 6e-3 else "BAD"
        ok = ok and err < 6e-3
        if rep == 0 or err >= 6e-3:
            print(f"{flag} v{v} M={M} N={N} K={K} K2={K2} kernel={choice} err={err:.2e}", flush=True)
lib.vlb_gemm_set_variant(3, 0)
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
