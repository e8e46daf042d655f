"""MX-fp8 GEMM: read-phase kernel (variant 0, rounds 1-2) vs the pipelined kernel (variant 1) on the decoder shapes -
bit-equality of the two outputs (same accumulation order) and HIP-event times, same process, tools build."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phantom_vlb_amd import ops
from phantom_vlb_amd._lib import lib
lib.vlb_gemm_mxfp8_set_variant.argtypes = [ctypes.c_int]; lib.vlb_gemm_mxfp8_set_variant.restype = None
dev = torch.device("cuda:0"); BF = torch.bfloat16


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


# variant word: bits 0-7 kernel (0 = round-2 read-phase kernel, 1 = pipelined, 2 / 3 = barrier S elsewhere), bits 8.. forced tile
# rows (0 = planner): 65537 = pipelined with 256-row tiles, 49153 = pipelined with 192-row tiles, 1 = pipelined, planner's choice
VARIANTS = [int(v, 0) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1").split(",")]
shapes = [("tiny", 77, 256, 128), ("k256", 300, 512, 256), ("k384", 300, 512, 384), ("halves", 3000, 3072, 256), ("qkv", 5861, 6144, 4096), ("o", 5861, 4096, 4096),
          ("gate_up", 5861, 28672, 4096), ("down", 5861, 4096, 14336), ("d_gu", 5861, 4096, 28672), ("d_down", 5861, 14336, 4096),
          ("wgrad gu", 28672, 4096, 5888), ("gate_up M=10240", 10240, 28672, 4096), ("sq8192", 8192, 8192, 8192)]
for name, M, N, K in shapes:
    torch.manual_seed(0)
    a = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
    r = torch.randn(M, N, device=dev).to(BF)
    aq, sa = ops.quantize_mxfp8(a); wq, sw = ops.quantize_mxfp8(w)
    outs, times = [], []
    for v in VARIANTS:
        lib.vlb_gemm_mxfp8_set_variant(v)
        o = torch.empty(M, N, dtype=BF, device=dev)
        ops.gemm_mxfp8(aq, sa, wq, sw, out=o)
        o_r = ops.gemm_mxfp8(aq, sa, wq, sw, residual=r)
        torch.cuda.synchronize()
        outs.append((o, o_r))
        times.append(t(lambda: ops.gemm_mxfp8(aq, sa, wq, sw, out=o)))
    same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
    fl = 2.0 * M * N * K
    print(f"{name:16s} M={M} N={N} K={K}: equal={same}  " + " | ".join(f"v{v} {tm:.3f} ms {fl / tm / 1e9:7.1f} TF" for v, tm in zip(VARIANTS, times))
          + f"  ({(times[0] / min(times[1:]) - 1) * 100:+.1f} %)", flush=True)
    del a, w, r, aq, wq, outs
