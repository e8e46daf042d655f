"""MX-fp8 quantisation passes at the full fine-tune's shapes: separate row-wise + transposed kernels vs the one-read dual kernel,
and the producers that emit the quantisation of their output (RMSNorm, SwiGLU) vs producer + quantise."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phantom_vlb_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
M = int(os.environ.get("VLB_ROWS", 5861))


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for R, C in ((M, 4096), (M, 6144), (M, 28672), (28672, 4096), (4096, 14336)):
    x = torch.randn(R, C, device=dev).to(BF)
    Rp = (R + 127) // 128 * 128
    q = torch.empty(R, C, dtype=torch.uint8, device=dev); s = torch.empty(R, C // 32, dtype=torch.uint8, device=dev)
    qt = torch.empty(C, Rp, dtype=torch.uint8, device=dev); st = torch.empty(C, Rp // 32, dtype=torch.uint8, device=dev)
    a = t(lambda: ops.quantize_mxfp8(x, q, s)); b = t(lambda: ops.transpose_quantize_mxfp8(x, qt, st, Rp))
    d = t(lambda: ops.quantize_dual_mxfp8(x, q, s, qt, st, Rp))
    gb = R * C / 1e9
    print(f"[{R}x{C}] row {a:6.1f} us ({3 * gb / a * 1e6 / 1e3:5.2f} TB/s)  transposed {b:6.1f} us ({3 * gb / b * 1e6 / 1e3:5.2f})  "
          f"dual {d:6.1f} us ({4 * gb / d * 1e6 / 1e3:5.2f})  vs separate {a + b:6.1f}", flush=True)
x = torch.randn(M, 4096, device=dev).to(BF); w = torch.ones(4096, device=dev).to(BF)
a = t(lambda: ops.rmsnorm(x, w, 1e-5)); y = ops.rmsnorm(x, w, 1e-5); b = t(lambda: ops.quantize_mxfp8(y)); c = t(lambda: ops.rmsnorm_mxfp8(x, w, 1e-5))
print(f"rmsnorm {a:.1f} + quantise {b:.1f} = {a + b:.1f} us | fused {c:.1f} us")
gu = torch.randn(M, 28672, device=dev).to(BF)
a = t(lambda: ops.swiglu(gu)); h = ops.swiglu(gu); b = t(lambda: ops.quantize_mxfp8(h)); c = t(lambda: ops.swiglu_mxfp8(gu))
print(f"swiglu {a:.1f} + quantise {b:.1f} = {a + b:.1f} us | fused {c:.1f} us")
