"""Short-K GEMMs of the CLIP tower (K = 1024: 16 K-tiles per output tile, a third of a one-workgroup-per-CU kernel is prologue + epilogue):
8-wave ping-pong kernel (product) vs four-wave kernel vs four-wave PERSISTENT stream (tools experiment, ABL bit 7: the next tile's first
K-tiles are in flight under the current epilogue).  Interleaved rounds, outputs compared."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phantom_vlb_amd import ops
from phantom_vlb_amd._lib import lib
lib.vlb_gemm_set_variant.argtypes = [ctypes.c_int, ctypes.c_int]; lib.vlb_gemm_set_variant.restype = None
lib.vlb_gemm_set_persist.argtypes = [ctypes.c_int]; lib.vlb_gemm_set_persist.restype = None
dev = torch.device("cuda:0"); BF = torch.bfloat16


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


arms = [("8-wave (product)", 3, 0), ("four-wave", 2, 0), ("four-wave persistent", 2, 1)]
shapes = [("vit_qkv", 20772, 3072, 1024), ("vit_out", 20772, 1024, 1024), ("vit_fc1", 20772, 4096, 1024), ("vit_fc2", 20772, 1024, 4096),
          ("gate_up K=4096", 5861, 28672, 4096)]
torch.manual_seed(0)
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev).to(BF)
    w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
    out = torch.empty(M, N, device=dev, dtype=BF)
    res, times = [], [[] for _ in arms]
    for _, v, p in arms:
        lib.vlb_gemm_set_variant(v, 0); lib.vlb_gemm_set_persist(p)
        out.zero_(); ops.gemm(a, w, out=out); torch.cuda.synchronize()
        res.append(out.clone())
    for rnd in range(5):
        order = range(len(arms)) if rnd % 2 == 0 else reversed(range(len(arms)))
        for i in order:
            lib.vlb_gemm_set_variant(arms[i][1], 0); lib.vlb_gemm_set_persist(arms[i][2])
            times[i].append(t(lambda: ops.gemm(a, w, out=out)))
    lib.vlb_gemm_set_variant(3, 0); lib.vlb_gemm_set_persist(0)
    fl = 2.0 * M * N * K
    print(f"{name:16s} [{M}x{N}x{K}] " + " | ".join(
        f"{arms[i][0]}: {min(times[i]):7.1f} us {fl / min(times[i]) / 1e6:6.0f} TF eq={torch.equal(res[0], res[i])}" for i in range(len(arms))), flush=True)
