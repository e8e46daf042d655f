"""A/B of the tile-row choice of the masked-pair dgrad GEMMs (LoRA backward of `o` and `down`) at the LoRA batch."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phantom_vlb_amd import ops
from phantom_vlb_amd._lib import lib
lib.vlb_gemm_set_variant.argtypes = [ctypes.c_int, ctypes.c_int]; lib.vlb_gemm_set_variant.restype = None
dev = torch.device("cuda:0"); BF = torch.bfloat16
M = 5861


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


u = torch.zeros(M, 64, dtype=BF, device=dev); u[:, :16] = torch.randn(M, 16, device=dev).to(BF)
for name, N, K, swi in (("down dgrad + swiglu bwd", 14336, 4096, True), ("o dgrad", 4096, 4096, False)):
    dy = torch.randn(M, K, device=dev).to(BF); wt = (torch.randn(N, K, device=dev) * 0.02).to(BF)
    At = torch.zeros(N, 64, dtype=BF, device=dev); At[:, :16] = (torch.randn(N, 16, device=dev) * 0.02).to(BF)
    gu = torch.randn(M, 2 * N, device=dev).to(BF) if swi else None
    out = torch.empty(M, 2 * N if swi else N, dtype=BF, device=dev)
    res = {}
    for rnd in range(3):
        for force, lab in ((0, "auto"), (3, "256"), (4, "192")):
            lib.vlb_gemm_set_variant(3, force)
            fn = (lambda: ops.gemm_masked_pair_swiglu_bwd(dy, wt, gu, u, At, 0.1, 1234, out=out)) if swi else \
                 (lambda: ops.gemm_masked_pair(dy, wt, u, At, 0.1, 1234, out=out))
            res[lab] = min(res.get(lab, 1e9), t(fn))
    lib.vlb_gemm_set_variant(3, 0)
    plain = t(lambda: ops.gemm(dy, wt))
    print(f"{name:26s} N={N} K={K}: " + "  ".join(f"{k} {v:.0f} us" for k, v in res.items()) + f"  | plain bf16 GEMM {plain:.0f} us", flush=True)
