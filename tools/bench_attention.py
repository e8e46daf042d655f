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
if hasattr(lib, "vlb_attn_set_fwd64"):
    # forward: 32 rows per wave (two workgroups per CU) vs 64 rows per wave (one), interleaved rounds; outputs must be bit-equal
    lib.vlb_attn_set_fwd64.argtypes = [ctypes.c_int]; lib.vlb_attn_set_fwd64.restype = None
    cases = [("decoder D=128 causal", qkv[:, :qd], qkv[:, qd:qd + kd], qkv[:, qd + kd:], B, S, Hq, Hkv, D, True, mask, 1),
             ("decoder D=128 causal B=5", None, None, None, 5, S, Hq, Hkv, D, True, None, 1)]
    T, SV, HV, DV = 36, 577, 16, 64
    vq = (torch.randn(T * SV, 3 * HV * DV, device=dev) * 0.5).bfloat16()
    cases.append(("ViT D=64 full", vq[:, :HV * DV], vq[:, HV * DV:2 * HV * DV], vq[:, 2 * HV * DV:], T, SV, HV, HV, DV, False, None, 2))
    for name, q_, k_, v_, b_, s_, hq_, hkv_, d_, causal, m_, bit in cases:
        if q_ is None:
            x_ = (torch.randn(b_ * s_, hq_ * d_ + 2 * hkv_ * d_, device=dev) * 0.5).bfloat16()
            q_, k_, v_ = x_[:, :hq_ * d_], x_[:, hq_ * d_:hq_ * d_ + hkv_ * d_], x_[:, hq_ * d_ + hkv_ * d_:]
        fn = lambda: ops.attention_fwd(q_, k_, v_, b_, s_, hq_, hkv_, d_, causal, d_ ** -0.5, key_mask=m_, need_lse=True)
        res, times = [], ([], [])
        for v in (0, bit):
            lib.vlb_attn_set_fwd64(v)
            o_, l_ = fn()
            res.append((o_.clone(), l_.clone()))
        for rnd in range(6):
            for idx in ((0, 1) if rnd % 2 == 0 else (1, 0)):
                lib.vlb_attn_set_fwd64((0, bit)[idx])
                times[idx].append(timeit(fn) * 1e6)
        lib.vlb_attn_set_fwd64(0)
        same = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        fl = 4.0 * b_ * hq_ * s_ * s_ * d_ / (2 if causal else 1)
        print(f"fwd {name:26s} equal={same}  32-row waves: min {min(times[0]):7.1f} us ({fl / min(times[0]) / 1e6:6.1f} TF) | 64-row waves: min {min(times[1]):7.1f} us "
              f"({fl / min(times[1]) / 1e6:6.1f} TF)  ({(min(times[0]) / min(times[1]) - 1) * 100:+.1f} %)", flush=True)
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
