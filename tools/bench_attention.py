"""Attention forward/backward timing at the decoder shape (B clips x S tokens, 32 q / 8 kv heads, D=128)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: E402,F401  (libvlb_tools.so: variant switches / ablations live only there)
from phantom_vlb_amd import ops
from phantom_vlb_amd._lib import lib

dev = torch.device("cuda:0")
B, S, Hq, Hkv, D = 3, 2048, 32, 8, 128
qd, kd = Hq * D, Hkv * D
qkv = (torch.randn(B * S, qd + 2 * kd, device=dev) * 0.5).bfloat16()
dout = torch.randn(B * S, qd, device=dev).bfloat16()
mask = torch.ones(B, S, dtype=torch.uint8, device=dev)


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


fwd_flops = 4.0 * B * Hq * S * S * D / 2
out, lse = ops.attention_fwd(qkv[:, :qd], qkv[:, qd:qd + kd], qkv[:, qd + kd:], B, S, Hq, Hkv, D, True, D ** -0.5, key_mask=mask, need_lse=True)
t = timeit(lambda: ops.attention_fwd(qkv[:, :qd], qkv[:, qd:qd + kd], qkv[:, qd + kd:], B, S, Hq, Hkv, D, True, D ** -0.5, key_mask=mask, need_lse=True))
print(f"fwd  {t*1e6:8.1f} us  {fwd_flops/t/1e12:6.1f} TF/s")
modes = [0]
if hasattr(lib, "vlb_attn_set_ablation"):
    lib.vlb_attn_set_ablation.argtypes = [ctypes.c_int]; lib.vlb_attn_set_ablation.restype = None
    modes = [0, 8, 4]           # 0: per-q-head dK/dV + dQ pass (default); 8: 8-wave per-kv-head dK/dV + dQ pass; 4: atomic dQ
for m in modes:
    lib.vlb_attn_set_ablation(m) if len(modes) > 1 else None
    acc = torch.empty(B * S, qd, dtype=torch.float32, device=dev) if m & 4 else None
    t = timeit(lambda: ops.attention_bwd(qkv, qd, kd, out, dout, lse, mask, B, S, Hq, Hkv, D, True, D ** -0.5, dq_acc=acc))
    print(f"bwd mode={m} {t*1e6:8.1f} us  {2.5*fwd_flops/t/1e12:6.1f} TF/s")
if len(modes) > 1:
    lib.vlb_attn_set_ablation(0)
