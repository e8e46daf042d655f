"""bf16 four-wave GEMM: k-step-split K loop (product, rounds 1-3) vs the row-split K loop (tools build, ABL bit 6) -
bit-equality of the outputs (same accumulation order per element) and HIP-event times for every call kind the LoRA step
makes: plain, residual, fused gate/up + SwiGLU + saved pre-activations (+ LoRA pair), masked-pair dgrad (+ SwiGLU backward)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phantom_vlb_amd import ops
from phantom_vlb_amd._lib import lib
MODE = os.environ.get("VLB_AB", "rowsplit")             # rowsplit | persist (the persistent-stream experiment, ABL bit 7) | wd (W-direct kernel on / off)
setter = {"persist": lib.vlb_gemm_set_persist, "wd": lib.vlb_gemm_set_wd, "streamk": lib.vlb_gemm_set_streamk}.get(MODE, lib.vlb_gemm_set_rowsplit)    # streamk: 0 = the round + tail plans, 1 = one stream-K launch (the product's choice)
setter.argtypes = [ctypes.c_int]; setter.restype = None
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


WPAD = int(os.environ.get("VLB_WPAD", 0))        # pad every weight's row stride by this many elements (non-power-of-two strides)


def padded(w):
    if not WPAD:
        return w
    buf = torch.zeros(w.shape[0], w.shape[1] + WPAD, dtype=w.dtype, device=w.device)
    buf[:, :w.shape[1]] = w
    return buf[:, :w.shape[1]]


torch.manual_seed(0)
x = torch.randn(M, 4096, device=dev).to(BF)
wqkv = padded((torch.randn(6144, 4096, device=dev) * 0.02).to(BF))
wo = padded((torch.randn(4096, 4096, device=dev) * 0.02).to(BF))
wgu = (torch.randn(28672, 4096, device=dev) * 0.02).to(BF)
wgu_il = padded(ops.interleave_gate_up(wgu[:14336], wgu[14336:]))
wgu = padded(wgu)
tl = torch.zeros(M, 64, dtype=BF, device=dev); tl[:, :32] = torch.randn(M, 32, device=dev).to(BF)
bp = torch.zeros(28672, 64, dtype=BF, device=dev); bp[:, :32] = (torch.randn(28672, 32, device=dev) * 0.02).to(BF)
dy = torch.randn(M, 4096, device=dev).to(BF); wt = padded((torch.randn(14336, 4096, device=dev) * 0.02).to(BF))
u = torch.zeros(M, 64, dtype=BF, device=dev); u[:, :16] = torch.randn(M, 16, device=dev).to(BF)
At = torch.zeros(14336, 64, dtype=BF, device=dev); At[:, :16] = (torch.randn(14336, 16, device=dev) * 0.02).to(BF)
gu = torch.randn(M, 28672, device=dev).to(BF)
hh = torch.randn(M, 14336, device=dev).to(BF); wd = padded((torch.randn(4096, 14336, device=dev) * 0.02).to(BF)); res = torch.randn(M, 4096, device=dev).to(BF)
dgu = torch.randn(M, 28672, device=dev).to(BF); wgut = padded((torch.randn(4096, 28672, device=dev) * 0.02).to(BF))


def outs(r):
    return r if isinstance(r, (tuple, list)) else (r,)


cases = {
    "qkv plain": lambda: ops.gemm(x, wqkv),
    "o + residual": lambda: ops.gemm(x, wo, residual=res),
    "gate/up plain": lambda: ops.gemm(x, wgu),
    "gate/up swiglu_save + LoRA pair": lambda: ops.gemm_swiglu_save(x, wgu_il, a2=tl, w2_il=bp),
    "down fwd + residual (K=14336)": lambda: ops.gemm(hh, wd, residual=res),
    "dgrad gate/up (K=28672)": lambda: ops.gemm(dgu, wgut),
    "dgrad down plain": lambda: ops.gemm(dy, wt),
    "qkv + LoRA pair": lambda: ops.gemm(x, wqkv, a2=tl, w2=bp[:6144]),
    "o + residual + LoRA pair": lambda: ops.gemm(x, wo, residual=res, a2=tl, w2=bp[:4096]),
    "dgrad qkv (K=6144)": lambda: ops.gemm(dgu[:, :6144], wgut[:, :6144]),
    "dgrad o masked pair": lambda: ops.gemm_masked_pair(dy, wo, u, At[:4096], 0.1, 4321),
    "dgrad down masked pair + swiglu bwd": lambda: ops.gemm_masked_pair_swiglu_bwd(dy, wt, gu, u, At, 0.1, 1234),
}
only = os.environ.get("VLB_CASE")
for name, fn in cases.items():
    if only and only not in name:
        continue
    res_ = []
    for v in (0, 1):
        setter(v)
        o = [q.clone() for q in outs(fn()) if torch.is_tensor(q)]
        torch.cuda.synchronize()
        res_.append(o)
    # interleaved rounds in one process (cdna guide rule 24): A B A B ..., min and median per arm
    times = ([], [])
    for rnd in range(int(os.environ.get("VLB_ROUNDS", 6))):
        for v in ((0, 1) if rnd % 2 == 0 else (1, 0)):
            setter(v)
            times[v].append(t(fn, 8))
    setter(0)          # back to the build's default
    same = all(torch.equal(a, b) for a, b in zip(res_[0], res_[1]))
    md = max(float((a.float() - b.float()).abs().max()) for a, b in zip(res_[0], res_[1]))
    mn = [min(x) for x in times]
    med = [sorted(x)[len(x) // 2] for x in times]
    print(f"{name:38s} equal={same} (max diff {md:.2e})  off: min {mn[0]:7.1f} med {med[0]:7.1f} us | {MODE}: min {mn[1]:7.1f} med {med[1]:7.1f} us  "
          f"(min {(mn[0] / mn[1] - 1) * 100:+.1f} %, med {(med[0] / med[1] - 1) * 100:+.1f} %)", flush=True)
    del res_
