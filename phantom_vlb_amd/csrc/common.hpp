// Shared device/host helpers for libvlb (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vlb.h"

typedef __bf16 bf16;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define VLB_WAVE 64

// ---- error plumbing (C-ABI: 0 ok, negative on error; message via vlb_last_error) ----
void vlb_set_error(const char* fmt, ...);

#define VLB_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      vlb_set_error(__VA_ARGS__);              \
      return VLB_ERR_INVALID;                  \
    }                                          \
  } while (0)

#define VLB_LAUNCH_CHECK()                                                   \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      vlb_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,           \
                    hipGetErrorString(e__));                                 \
      return VLB_ERR_LAUNCH;                                                 \
    }                                                                        \
  } while (0)

// Every entry point converts its stream argument right before launching: clear any stale (sticky)
// error another HIP user of this thread left behind, so VLB_LAUNCH_CHECK reports only our own launch.
static inline hipStream_t as_stream(void* s) {
  (void)hipGetLastError();
  return reinterpret_cast<hipStream_t>(s);
}

// ---- device helpers ----
__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x multiple of 64, <= 1024.  `red` is >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// 1 / (1 + e^-x) through v_rcp_f32 (1 ulp) instead of an IEEE division (v_div_scale / v_div_fmas / v_div_fixup, ~10 instructions):
// these run once per output element in the GEMM epilogues (SwiGLU forward / backward, quick-GELU), where with one wave per
// SIMD every VALU instruction is on the tile's serial tail; the result is rounded to bf16 right after.
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
__device__ __forceinline__ float quick_gelu_f(float x) { return x * sigmoid_f(1.702f * x); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

// 16-byte epilogue stores.  In the accumulator layout a lane (fr, fq) owns columns 4fq..4fq+3 of a 16-column fragment,
// i.e. 8 bytes of bf16: the four lanes of a row would each store 8 bytes per fragment.  For two ADJACENT fragments j, j+1
// (a = this lane's piece of j, b = of j+1) one v_permlane16_swap per dword hands the odd-fq lanes' pieces of j to their
// even neighbours and the even lanes' pieces of j+1 to the odd ones: an even lane then holds columns 4fq..4fq+7 of
// fragment j, an odd lane columns 4(fq-1)..4(fq-1)+7 of fragment j+1 - one 16-byte store each instead of two 8-byte
// ones (half the store instructions of the serial tile tail, 64 contiguous bytes per row per instruction).
// row = &C[m][0], n = the column of this lane's piece of fragment j (16-byte aligned for even fq).
// MX (OCP microscaling) block quantisation helpers shared by gemm_fp8.hip and the producer kernels that emit fp8 next to bf16:
// shared exponent of a 32-element block = ceil(log2(amax / 448)) clamped to [-127, 127]; scale byte = E + 127.
__device__ __forceinline__ int mx_shared_exp(float amax) {
  int E = -127;
  if (amax > 0.f) {
    int e; const float f = frexpf(amax / 448.f, &e);        // amax/448 = f * 2^e, f in [0.5,1)
    E = (f == 0.5f) ? e - 1 : e;
    E = max(-127, min(127, E));
  }
  return E;
}
// four floats -> four OCP e4m3 bytes (v_cvt_pk_fp8_f32: round-to-nearest-even; the scaled block never exceeds 448)
__device__ __forceinline__ uint32_t mx_pack4(float a, float b, float c, float d, float inv) {
  int pk = 0;
  pk = __builtin_amdgcn_cvt_pk_fp8_f32(a * inv, b * inv, pk, false);
  pk = __builtin_amdgcn_cvt_pk_fp8_f32(c * inv, d * inv, pk, true);
  return (uint32_t)pk;
}
// A lane holds 8 consecutive (bf16-valued) elements of a row and lanes 4q..4q+3 hold one 32-element block: quantise them
// exactly as quantize_mxfp8_kernel does - 8 bytes per lane, the block's scale byte stored by the lane with (lane & 3) == 0.
__device__ __forceinline__ void mx_quantize_lane8(const float (&v)[8], uint8_t* qrow, uint8_t* srow, int c, int lane) {
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(v[i]));
  amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
  amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
  const int E = mx_shared_exp(amax);
  const float inv = exp2f((float)-E);
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
  *reinterpret_cast<u32x2_*>(qrow + c) = u32x2_{mx_pack4(v[0], v[1], v[2], v[3], inv), mx_pack4(v[4], v[5], v[6], v[7], inv)};
  if ((lane & 3) == 0) srow[c >> 5] = (uint8_t)(E + 127);
}
// The read side of the same pairing: ONE 16-byte load per lane for this lane's pieces of fragments j and j+1 of a bf16 row
// (even fq: fragment j's columns of lanes fq, fq+1; odd fq: fragment j+1's of fq-1, fq), then the same two lane-row swaps
// hand every lane its own two 4-column pieces.  row + n must be 16-byte aligned for even fq (as for store_pair16).
__device__ __forceinline__ void load_pair16(const bf16* row, int n, int fq, bf16x4& a, bf16x4& b) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 l = *reinterpret_cast<const u32x4*>(row + n + ((fq & 1) ? 12 : 0));
  const auto r0 = __builtin_amdgcn_permlane16_swap(l[0], l[2], false, false);
  const auto r1 = __builtin_amdgcn_permlane16_swap(l[1], l[3], false, false);
  a = __builtin_bit_cast(bf16x4, u32x2{r0[0], r1[0]});
  b = __builtin_bit_cast(bf16x4, u32x2{r0[1], r1[1]});
}
__device__ __forceinline__ void store_pair16(bf16* row, int n, bf16x4 a, bf16x4 b, int fq) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
  const auto r0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
  const auto r1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
  const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
  *reinterpret_cast<u32x4*>(row + n + ((fq & 1) ? 12 : 0)) = o;
}
