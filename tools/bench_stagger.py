"""Timing experiment (tools build): does de-synchronising the CUs shorten the GEMM calls whose epilogues move a lot of memory?
Half of the first-round workgroups start `ticks` x 10 ns late (vlb_gemm_set_stagger); every call still does all its work."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phantom_vlb_amd import ops
from phantom_vlb_amd._lib import lib
lib.vlb_gemm_set_stagger.argtypes = [ctypes.c_int]; lib.vlb_gemm_set_stagger.restype = None
dev = torch.device("cuda:0"); BF = torch.bfloat16
M = int(os.environ.get("VLB_ROWS", 5861))


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


x = torch.randn(M, 4096, device=dev).to(BF)
wgu = (torch.randn(28672, 4096, device=dev) * 0.02).to(BF)
wgu_il = ops.interleave_gate_up(wgu[:14336], wgu[14336:])
tl = torch.zeros(M, 64, dtype=BF, device=dev); tl[:, :32] = torch.randn(M, 32, device=dev).to(BF)
bp = torch.zeros(28672, 64, dtype=BF, device=dev); bp[:, :32] = (torch.randn(28672, 32, device=dev) * 0.02).to(BF)
dy = torch.randn(M, 4096, device=dev).to(BF); wt = (torch.randn(14336, 4096, device=dev) * 0.02).to(BF)
u = torch.zeros(M, 64, dtype=BF, device=dev); u[:, :16] = torch.randn(M, 16, device=dev).to(BF)
At = torch.zeros(14336, 64, dtype=BF, device=dev); At[:, :16] = (torch.randn(14336, 16, device=dev) * 0.02).to(BF)
gu = torch.randn(M, 28672, device=dev).to(BF)
out2 = torch.empty(M, 28672, dtype=BF, device=dev)
hh = torch.randn(M, 14336, device=dev).to(BF); wd = (torch.randn(4096, 14336, device=dev) * 0.02).to(BF); res = torch.randn(M, 4096, device=dev).to(BF)
cases = {
    "gate/up plain GEMM": lambda: ops.gemm(x, wgu),
    "gate/up swiglu_save (+LoRA pair)": lambda: ops.gemm_swiglu_save(x, wgu_il, a2=tl, w2_il=bp),
    "down dgrad masked + swiglu bwd": lambda: ops.gemm_masked_pair_swiglu_bwd(dy, wt, gu, u, At, 0.1, 1234, out=out2),
    "down fwd (+residual)": lambda: ops.gemm(hh, wd, residual=res),
}
ticks = [0, 300, 600, 1000, 1500, 2500, 4000]
for name, fn in cases.items():
    best = {k: 1e9 for k in ticks}
    for rnd in range(3):
        for k in ticks:
            lib.vlb_gemm_set_stagger(k)
            best[k] = min(best[k], t(fn))
    lib.vlb_gemm_set_stagger(0)
    print(f"{name:34s} M={M}: " + "  ".join(f"{k * 10 / 1000:.0f}us:{v:.0f}" for k, v in best.items()), flush=True)
