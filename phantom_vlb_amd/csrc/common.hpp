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

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float quick_gelu_f(float x) { return x * sigmoid_f(1.702f * x); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
