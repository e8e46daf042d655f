"""A/B of the GEMM kernel variants (interleaved rounds in one process, random data)."""
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


def main():
    dev = torch.device("cuda:0")
    variants = [int(v, 0) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,2,3,4,5".split(","))]
    shapes = [("qkv", 10240, 6144, 4096), ("o", 10240, 4096, 4096), ("gate_up", 10240, 28672, 4096),
              ("down", 10240, 4096, 14336), ("sq8192", 8192, 8192, 8192), ("vit_fc1", 34620, 4096, 1024)]
    if os.environ.get("VLB_SHAPES") == "lora":          # the LoRA batch (M = 5861 packed rows): every GEMM has a re-cut tail
        shapes = [("qkv", 5861, 6144, 4096), ("o", 5861, 4096, 4096), ("gate_up", 5861, 28672, 4096), ("down", 5861, 4096, 14336),
                  ("d_gu", 5861, 4096, 28672), ("d_down", 5861, 14336, 4096)]
    if os.environ.get("VLB_SHAPES") == "frozen":        # the frozen batch (M = 9447 packed rows)
        shapes = [("qkv", 9447, 6144, 4096), ("o", 9447, 4096, 4096), ("gate_up", 9447, 28672, 4096), ("down", 9447, 4096, 14336)]
    if os.environ.get("VLB_SHAPES") == "vit":           # the CLIP tower at the LoRA batch: 3 clips x 12 frames x 577 tokens, width 1024
        shapes = [("vit_qkv", 20772, 3072, 1024), ("vit_out", 20772, 1024, 1024), ("vit_fc1", 20772, 4096, 1024), ("vit_fc2", 20772, 1024, 4096),
                  ("conn_1x1", 20736, 4096, 4096), ("conn_in", 20736, 4096, 1024)]
    if os.environ.get("VLB_SHAPES", "").startswith("rows="):   # VLB_SHAPES=rows=31200: the decoder projections at another row count
        M = int(os.environ["VLB_SHAPES"][5:])
        shapes = [("qkv", M, 6144, 4096), ("o", M, 4096, 4096), ("gate_up", M, 28672, 4096), ("down", M, 4096, 14336)]
    for name, M, N, K in shapes:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ref = a[-256:].float() @ w.float().t()          # the last rows: they live in the tail launch
        best = {v: 1e9 for v in variants}
        errs = {}
        for rnd in range(4):
            for v in variants:
                lib.vlb_gemm_set_variant(v, 0)
                out.zero_()
                for _ in range(2):
                    ops.gemm(a, w, out=out)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.gemm(a, w, out=out)
                e1.record()
                torch.cuda.synchronize()
                best[v] = min(best[v], e0.elapsed_time(e1) / 10)
                errs[v] = float((out[-256:].float() - ref).abs().max() / ref.abs().max())
        print(f"{name:8s} M={M} N={N} K={K}: " + "  ".join(
            f"v{v}: {best[v]:.3f}ms {2.0 * M * N * K / best[v] / 1e9:7.1f}TF err={errs[v]:.0e}" for v in variants), flush=True)
    lib.vlb_gemm_set_variant(3, 0)


if __name__ == "__main__":
    main()
