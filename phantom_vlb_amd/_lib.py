"""ctypes binding of libvlb.so (the C-ABI declared in include/vlb.h).

The product path has NO fallback: if the HIP library is missing or an entry point is absent,
import fails loudly.  Nothing here imports ``oracle/``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_void_p

# torch FIRST: its wheel bundles its own libamdhip64; libvlb.so must bind to that already-loaded HIP
# runtime (same SONAME) instead of pulling a second copy from /opt/rocm, or the two runtimes fight
# over the device ("no ROCm-capable device is detected" on the first libvlb launch).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# VLB_LIB: tools/*.py point this at libvlb_tools.so (the -DVLB_TOOLS build with kernel-variant switches and
# timing-only ablations) for A/B runs.  bench.py and the tests refuse anything but the in-tree product library.
LIB_PATH = os.environ.get("VLB_LIB") or os.path.join(_HERE, "libvlb.so")
IS_PRODUCT_LIB = os.path.abspath(LIB_PATH) == os.path.join(_HERE, "libvlb.so")

P, I, F, L = c_void_p, c_int, c_float, c_int64

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/vlb.h one to one
SIGNATURES = {
    "vlb_abi_version": [],
    "vlb_last_error": [],
    "vlb_gemm_bf16": [P, I, P, I, P, I, I, I, I, P, P, I, I, P, I, P, I, I, P],
    "vlb_gemm_bf16_ws": [P, I, P, I, P, I, I, I, I, P, P, I, I, P, I, P, I, I, P, L, P],
    "vlb_gemm_swiglu_save": [P, I, P, I, P, I, P, I, I, I, I, P, I, P, I, I, P, L, P],
    "vlb_gemm_workspace_bytes": [],
    "vlb_gemm_plan": [I, I, I, I, I],
    "vlb_gemm_kernel_choice": [I, I, I, I],
    "vlb_transpose_bf16": [P, P, I, I, P],
    "vlb_attention_fwd": [P, I, P, I, P, I, P, I, P, P, I, I, I, I, I, I, F, P, P],
    "vlb_attention_bwd": [P, I, P, I, P, I, P, I, P, I, P, P, P, I, P, I, P, I, P, P, I, I, I, I, I, I, F, P, I, P],
    "vlb_rmsnorm_fwd": [P, P, P, I, I, F, P],
    "vlb_rmsnorm_bwd": [P, P, P, P, P, I, I, F, P],
    "vlb_layernorm_fwd": [P, P, P, P, P, I, I, F, I, P],
    "vlb_rope_inplace": [P, I, P, P, I, I, I, I, I, P, P],
    "vlb_swiglu_fwd": [P, P, I, I, P],
    "vlb_swiglu_bwd": [P, P, P, I, I, P],
    "vlb_add_bf16": [P, P, P, L, P],
    "vlb_patchify": [P, P, I, I, I, I, I, P],
    "vlb_vit_assemble": [P, P, P, P, I, I, I, P],
    "vlb_drop_cls": [P, P, I, I, I, P],
    "vlb_dwconv3x3": [P, P, P, I, I, I, I, P],
    "vlb_se_pool": [P, P, I, I, I, P],
    "vlb_se_scale": [P, P, P, I, I, I, P],
    "vlb_im2col3d_k2s2p1": [P, P, I, I, I, I, I, P],
    "vlb_splice_embed": [P, P, P, P, P, P, I, I, I, I, L, I, P, P, P],
    "vlb_weight_mask": [P, P, P, P, I, I, I, I, I, I, P],
    "vlb_head_partial_rows": [I],
    "vlb_head_ws_floats": [I, I, I, I],
    "vlb_head_fwd": [P] * 19 + [I, I, I, I, F, F, P, P],
    "vlb_head_bwd": [P] * 24 + [I, I, I, I, F, F, F, F, P, I, P],
    "vlb_gemm_bf16_masked_pair": [P, I, P, I, P, I, I, I, I, P, I, P, I, F, ctypes.c_uint32, P],
    "vlb_gemm_masked_pair_swiglu_bwd": [P, I, P, I, P, I, P, I, I, I, I, P, I, P, I, F, ctypes.c_uint32, P, L, P],
    "vlb_gemm_bf16_masked_pair_ws": [P, I, P, I, P, I, I, I, I, P, I, P, I, F, ctypes.c_uint32, P, L, P],
    "vlb_wgrad_u_ws_floats": [I, I],
    "vlb_wgrad_skinny_u": [P, I, P, I, P, P, I, I, F, F, P, F, P, I, P, P],
    "vlb_wgrad_skinny_u_multi": [P, I, P, I, I, I, P, P, P, P, F, F, F, P, I, P, P],
    "vlb_transpose16_scatter": [P, I, P],
    "vlb_wgrad_splits": [I],
    "vlb_wgrad_skinny": [P, I, P, I, P, P, I, I, I, F, F, F, P, P],
    "vlb_lora_down": [P, I, P, P, I, I, I, I, F, F, P, P],
    "vlb_lora_dx_masked": [P, I, P, I, P, I, I, I, I, F, P, P],
    "vlb_sumsq_ws_floats": [],
    "vlb_grad_sumsq": [P, L, P, P, P],
    "vlb_adamw_step": [P, P, P, P, P, L, F, F, F, F, F, I, P, F, P],
    "vlb_dropout_keep_scale": [P, L, F, ctypes.c_uint32, P],
    "vlb_transpose_pad": [P, I, P, I, I, I, I, P],
    "vlb_norm_bwd_ws_floats": [I, I],
    "vlb_colsum_ws_floats": [I, I],
    "vlb_rmsnorm_bwd_dw": [P, P, P, P, I, I, F, P],
    "vlb_rmsnorm_bwd_full_ws_floats": [I, I],
    "vlb_rmsnorm_bwd_full": [P, P, P, P, P, P, P, I, I, F, P],
    "vlb_layernorm_bwd": [P, P, P, P, P, P, P, P, P, P, I, I, F, I, P],
    "vlb_act_fwd": [P, P, L, I, P],
    "vlb_act_bwd": [P, P, P, L, I, P],
    "vlb_colsum": [P, I, P, P, I, I, P],
    "vlb_embed_grad": [P, I, P, P, P, I, P, I, P],
    "vlb_grad_sumsq_bf16": [P, L, P, P, P],
    "vlb_adamw_step_g16": [P, P, P, P, P, L, F, F, F, F, F, I, P, F, P],
    "vlb_dwconv3x3_bwd_w_ws_floats": [I, I, I],
    "vlb_dwconv3x3_bwd_w": [P, P, P, P, I, I, I, I, P],
    "vlb_se_bwd_gate": [P, P, P, P, I, I, I, P],
    "vlb_se_bwd_x": [P, P, P, P, I, I, I, P],
    "vlb_col2im3d_k2s2p1": [P, P, I, I, I, I, I, P],
    "vlb_quantize_mxfp8": [P, I, P, I, P, I, I, I, P],
    "vlb_transpose_quantize_mxfp8": [P, I, P, I, P, I, I, I, I, P],
    "vlb_quantize_dual_mxfp8": [P, I, P, I, P, I, P, I, P, I, I, I, I, P],
    "vlb_rmsnorm_fwd_mxfp8": [P, P, P, P, I, P, I, I, I, F, P],
    "vlb_swiglu_fwd_mxfp8": [P, P, P, I, P, I, I, I, P],
    "vlb_gemm_mxfp8": [P, I, P, I, P, I, P, I, P, I, I, I, I, P, I, P],
    "vlb_comm_unique_id": [P],
    "vlb_comm_init": [I, I, P, ctypes.POINTER(c_void_p)],
    "vlb_comm_destroy": [P],
    "vlb_comm_loopback_stage_bytes": [I],
    "vlb_comm_init_loopback": [I, P, L, ctypes.POINTER(c_void_p)],
    "vlb_comm_rank": [P],
    "vlb_comm_world": [P],
    "vlb_allgather_direct": [P, P, P, L, P],
    "vlb_reducescatter_stage_floats": [L, I],
    "vlb_reducescatter_direct": [P, P, P, L, P, P],
    "vlb_reducescatter_direct_bf16": [P, P, P, L, P, P],
    "vlb_reduce_slices": [P, P, L, I, I, P],
    "vlb_profile_marker": [P],
    "vlb_hrf_pool_ws_floats": [I, I],
    "vlb_hrf_pool": [P, I, P, P, P, I, I, I, P],
    "vlb_ridge_ws_floats": [I],
    "vlb_ridge_fwd": [P, P, P, P, P, P, I, I, I, F, P],
    "vlb_allreduce_scalar": [P, P, I, P],
    "vlb_cast_f32_to_bf16": [P, P, L, P],
    "vlb_cast_bf16_to_f32": [P, P, L, P],
}
_RESTYPES = {"vlb_last_error": c_char_p, "vlb_hrf_pool_ws_floats": c_int64, "vlb_ridge_ws_floats": c_int64, "vlb_head_ws_floats": c_int64, "vlb_wgrad_u_ws_floats": c_int64,
             "vlb_gemm_workspace_bytes": c_int64, "vlb_reducescatter_stage_floats": c_int64, "vlb_comm_loopback_stage_bytes": c_int64,
             "vlb_norm_bwd_ws_floats": c_int64, "vlb_rmsnorm_bwd_full_ws_floats": c_int64, "vlb_colsum_ws_floats": c_int64, "vlb_dwconv3x3_bwd_w_ws_floats": c_int64}


class VlbError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C phantom_vlb_amd/csrc`.  phantom_vlb_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise ImportError(f"libvlb.so does not export {name}: rebuild it") from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    if lib.vlb_abi_version() != 2:
        raise ImportError("libvlb.so ABI version mismatch")
    return lib


lib = _load()


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise VlbError(f"{what} failed ({rc}): {lib.vlb_last_error().decode()}")
