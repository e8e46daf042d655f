"""HBM-side efficiency of the skinny LoRA kernels at the 7B / B=3 shapes (algorithmic bytes / time)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phantom_vlb_amd._lib import lib
from phantom_vlb_amd.lora import lora_down, lora_dx_masked, wgrad_skinny, PAD

dev = torch.device("cuda:0")
BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 5861


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
    return e0.elapsed_time(e1) / n * 1e-3


seeds3 = [11, 22, 33]
for K, G, p in [(4096, 3, 0.1), (4096, 3, 0.0), (4096, 2, 0.1), (4096, 2, 0.0), (4096, 1, 0.1), (4096, 1, 0.0), (14336, 1, 0.1), (14336, 1, 0.0), (1024, 1, 0.0)]:
    R = 16 * G
    x = torch.randn(M, K, device=dev).to(BF)
    A = (torch.randn(R, K, device=dev) * 0.02).to(BF)
    t = torch.zeros(M, PAD, dtype=BF, device=dev)
    dt = timeit(lambda: lora_down(x, A, R, 2.0, p, seeds3[:G], t))
    byt = M * K * 2 + R * K * 2 + M * R * 2
    print(f"lora_down  K={K:5d} G={G} p={p}: {dt*1e6:7.1f} us  {byt/dt/1e12:5.2f} TB/s", flush=True)
for K, G, p in [(4096, 3, 0.1), (4096, 3, 0.0), (4096, 2, 0.1), (4096, 2, 0.0), (4096, 1, 0.1), (4096, 1, 0.0), (14336, 1, 0.1), (14336, 1, 0.0)]:
    R = 16 * G
    u = torch.randn(M, PAD, device=dev).to(BF)
    At = torch.zeros(K, PAD, dtype=BF, device=dev)
    At[:, :R] = (torch.randn(K, R, device=dev) * 0.02).to(BF)
    dx = torch.randn(M, K, device=dev).to(BF)
    dt = timeit(lambda: lora_dx_masked(u, At, dx, R, p, seeds3[:G]))
    byt = 2 * M * K * 2 + M * R * 2
    print(f"lora_dx    K={K:5d} G={G} p={p}: {dt*1e6:7.1f} us  {byt/dt/1e12:5.2f} TB/s", flush=True)
for K, N, p in [(4096, 48, 0.1), (4096, 48, 0.0), (4096, 32, 0.1), (4096, 32, 0.0), (4096, 16, 0.1), (4096, 16, 0.0), (14336, 16, 0.1), (14336, 16, 0.0), (1024, 16, 0.0)]:
    Gm = torch.randn(M, PAD, device=dev).to(BF)
    X = torch.randn(M, K, device=dev).to(BF)
    dW = torch.zeros(N, K, dtype=torch.float32, device=dev)
    ws = torch.empty(lib.vlb_wgrad_splits(M) * 48 * K, dtype=torch.float32, device=dev)
    dt = timeit(lambda: wgrad_skinny(Gm, X, dW, ws, N, p=p, seeds=seeds3[:N // 16]))
    byt = M * K * 2 + M * N * 2 + N * K * 4
    print(f"wgrad      K={K:5d} N={N} p={p}: {dt*1e6:7.1f} us  {byt/dt/1e12:5.2f} TB/s", flush=True)
# dB^T + u fused (one pass over dy) vs the lora_down + wgrad pair
from phantom_vlb_amd.lora import wgrad_skinny_u
for K in (4096, 14336, 1024):
    t_ = torch.randn(M, PAD, device=dev).to(BF)
    dy = torch.randn(M, K, device=dev).to(BF)
    Bt = (torch.randn(16, K, device=dev) * 0.02).to(BF)
    dW = torch.zeros(16, K, dtype=torch.float32, device=dev)
    ws = torch.empty(lib.vlb_wgrad_splits(M) * 48 * K, dtype=torch.float32, device=dev)
    uws = torch.empty(lib.vlb_wgrad_u_ws_floats(M, K), dtype=torch.float32, device=dev)
    u = torch.zeros(M, PAD, dtype=BF, device=dev)
    def pair():
        lora_down(dy, Bt, 16, 2.0, 0.0, None, u)
        wgrad_skinny(t_, dy, dW, ws, 16)
    ta = timeit(pair)
    tb = timeit(lambda: wgrad_skinny_u(t_, dy, dW, ws, Bt, 2.0, u, uws))
    print(f"dB+u      K={K:5d}: separate {ta*1e6:7.1f} us  fused {tb*1e6:7.1f} us", flush=True)
