// LoRA adapter kernels (peft semantics: y = x W^T + s * B(A(dropout_p(x))), only A and B train).
//
// The base GEMM carries the adapter through vlb_gemm_bf16's second operand pair; what is left are
// the skinny rank-r products, which are HBM-bound (one pass over an [M,K] activation each):
//   vlb_lora_down       t[M,R]  = scale * (keep(x)/(1-p)) . A^T          (R = 16 * projections sharing x)
//   vlb_lora_dx_masked  dx[M,K] += keep/(1-p) * (u . A)                   (backward through dropout)
//   vlb_wgrad_skinny    dW[N,K] = alpha * G^T . (keep(X)/(1-p)) + beta*dW (dA and dB^T)
// Dropout masks are counter-based (a 32-bit hash of (seed, row, column pair)), so forward and backward
// regenerate the same mask and nothing is stored; each 16-rank group has its own seed, like peft's
// independent Dropout modules.
#include "common.hpp"

namespace {

__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// 16 random bits for element (m, k) of an [M,K] activation under `seed` (K even)
__device__ __forceinline__ uint32_t drop_bits_pair(uint32_t seed, int64_t m, int K, int kpair) {
  const uint64_t c = (uint64_t)m * (uint64_t)(K >> 1) + (uint64_t)kpair;
  return lowbias32((uint32_t)c ^ lowbias32((uint32_t)(c >> 32) + seed));
}
// keep flags of 8 consecutive columns starting at k0 (k0 % 8 == 0)
__device__ __forceinline__ void keep8(uint32_t seed, int64_t m, int K, int k0, uint32_t thresh, bool (&keep)[8]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t h = drop_bits_pair(seed, m, K, (k0 >> 1) + q);
    keep[2 * q] = (h & 0xffffu) >= thresh;
    keep[2 * q + 1] = (h >> 16) >= thresh;
  }
}

struct Seeds { uint32_t s[8]; };

// ---------------------------------------------------------------- t = scale * drop(x) . A^T
// one wave = 16 rows of x; MFMA rows = adapter ranks, MFMA cols = rows of x.
template <int G>   // number of 16-rank groups (projections sharing this x)
__global__ __launch_bounds__(256) void lora_down_kernel(const bf16* __restrict__ x, int ldx, const bf16* __restrict__ A,
                                                        bf16* __restrict__ t, int ldt, int M, int K, float scale,
                                                        uint32_t thresh, Seeds seeds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = (blockIdx.x * 4 + wave) * 16;
  if (m0 >= M) return;
  const int fr = lane & 15, fq = lane >> 4;
  const int m = min(m0 + fr, M - 1);
  f32x4 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16* xr = x + (int64_t)m * ldx + fq * 8;
  for (int k0 = 0; k0 < K; k0 += 32) {
    const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xr + k0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(A + (int64_t)(16 * g + fr) * K + k0 + fq * 8);
      bf16x8 xm = xf;
      if (thresh != 0) {
        bool keep[8];
        keep8(seeds.s[g], m, K, k0 + fq * 8, thresh, keep);
#pragma unroll
        for (int j = 0; j < 8; ++j) xm[j] = keep[j] ? xf[j] : (bf16)0.f;
      }
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xm, acc[g], 0, 0, 0);
    }
  }
  if (m0 + fr < M) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[g][e] * scale);
      *reinterpret_cast<bf16x4*>(t + (int64_t)(m0 + fr) * ldt + 16 * g + fq * 4) = o;
    }
  }
}

// ---------------------------------------------------------------- dx += keep/(1-p) * (u . A)
// At[K][R] is the transposed adapter (row = input column).  One wave = 16 rows x 16 columns per MFMA;
// a wave walks 16 rows x 256 columns.
template <int G>
__global__ __launch_bounds__(256) void lora_dx_kernel(const bf16* __restrict__ u, int ldu, const bf16* __restrict__ At, int ldat,
                                                      bf16* __restrict__ dx, int lddx, int M, int K, float inv_keep,
                                                      uint32_t thresh, Seeds seeds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = blockIdx.y * 16;
  const int kbase = (blockIdx.x * 4 + wave) * 256;
  if (kbase >= K) return;
  const int fr = lane & 15, fq = lane >> 4;
  const int m = min(m0 + fr, M - 1);
  // B operand: u[m][16g + 8fq .. +8] for fq < 2, zeros for the padded half of the k=32 step
  bf16x8 uf[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    uf[g] = bf16x8{};
    if (fq < 2) uf[g] = *reinterpret_cast<const bf16x8*>(u + (int64_t)m * ldu + 16 * g + 8 * fq);
  }
  for (int kc = 0; kc < 256 && kbase + kc < K; kc += 16) {
    const int kcol = kbase + kc;
    f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; ++g) {
      bf16x8 af = bf16x8{};
      if (fq < 2) af = *reinterpret_cast<const bf16x8*>(At + (int64_t)(kcol + fr) * ldat + 16 * g + 8 * fq);
      f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, uf[g], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      // lane holds columns kcol + 4fq + {0..3} of row m0 + fr
      if (thresh != 0) {
        const int kk = kcol + 4 * fq;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const uint32_t h = drop_bits_pair(seeds.s[g], m, K, (kk >> 1) + q);
          if ((h & 0xffffu) < thresh) d[2 * q] = 0.f;
          if ((h >> 16) < thresh) d[2 * q + 1] = 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) sum[e] += d[e] * inv_keep;
    }
    if (m0 + fr < M) {
      bf16* p = dx + (int64_t)(m0 + fr) * lddx + kcol + 4 * fq;
      const bf16x4 old = *reinterpret_cast<const bf16x4*>(p);
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)((float)old[e] + sum[e]);
      *reinterpret_cast<bf16x4*>(p) = o;
    }
  }
}

// ---------------------------------------------------------------- skinny wgrad
constexpr int WG_N = 16;        // output rows per pass
constexpr int WG_ROWS = 128;    // rows of M per split (minimum)

// grid (ceil(K/512), splits), block 256.  ws[split][N][K] partial slabs.
__global__ __launch_bounds__(256) void wgrad_partial_kernel(const bf16* __restrict__ G, int ldg, const bf16* __restrict__ X,
                                                            int ldx, float* __restrict__ ws, int M, int N, int K, int n0,
                                                            int rows_per_split, uint32_t thresh, uint32_t seed) {
  __shared__ float red[WG_N][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 512 + lane * 8;
  const int r0 = blockIdx.y * rows_per_split, r1 = min(M, r0 + rows_per_split);
  const int nn = min(WG_N, N - n0);
  float acc[WG_N][8];
#pragma unroll
  for (int i = 0; i < WG_N; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
  if (c < K) {
    for (int m = r0 + wave; m < r1; m += 4) {
      const bf16x8 xv = *reinterpret_cast<const bf16x8*>(X + (int64_t)m * ldx + c);
      float xf[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) xf[j] = (float)xv[j];
      if (thresh != 0) {
        bool keep[8];
        keep8(seed, m, K, c, thresh, keep);
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[j] = keep[j] ? xf[j] : 0.f;
      }
      const bf16* gr = G + (int64_t)m * ldg + n0;
#pragma unroll
      for (int i = 0; i < WG_N; ++i) {
        if (i < nn) {
          const float gv = (float)gr[i];
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[i][j] += gv * xf[j];
        }
      }
    }
  }
  // sum the 4 waves in a fixed order
  for (int w = 1; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < WG_N; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) red[i][lane * 8 + j] = acc[i][j];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < WG_N; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] += red[i][lane * 8 + j];
    }
  }
  if (wave == 0 && c < K) {
#pragma unroll
    for (int i = 0; i < WG_N; ++i) {
      if (i < nn) {
        float* o = ws + ((int64_t)blockIdx.y * N + n0 + i) * K + c;
        *reinterpret_cast<f32x4*>(o) = f32x4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
        *reinterpret_cast<f32x4*>(o + 4) = f32x4{acc[i][4], acc[i][5], acc[i][6], acc[i][7]};
      }
    }
  }
}
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dW, int splits, int64_t nk, float alpha,
                                    float beta) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nk; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += ws[(int64_t)k * nk + i];
    dW[i] = alpha * s + (beta != 0.f ? beta * dW[i] : 0.f);
  }
}

inline uint32_t thresh16(float p) {
  if (p <= 0.f) return 0;
  uint32_t t = (uint32_t)(p * 65536.f + 0.5f);
  return t > 65535u ? 65535u : t;
}
}  // namespace

extern "C" int vlb_wgrad_splits(int M) {
  int s = (M + WG_ROWS - 1) / WG_ROWS;
  return s < 1 ? 1 : (s > 32 ? 32 : s);
}

extern "C" int vlb_wgrad_skinny(const void* G, int ldg, const void* X, int ldx, float* dW, float* ws, int M, int N, int K,
                                float alpha, float beta, float drop_p, uint32_t drop_seed, void* stream) {
  VLB_REQUIRE(G && X && dW && ws, "wgrad_skinny: null operand");
  VLB_REQUIRE(M > 0 && N > 0 && N <= 64 && K > 0 && K % 8 == 0 && ldx % 8 == 0, "wgrad_skinny: bad shape M=%d N=%d K=%d", M, N, K);
  VLB_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "wgrad_skinny: bad dropout p");
  hipStream_t st = as_stream(stream);
  const int splits = vlb_wgrad_splits(M);
  const int rps = (M + splits - 1) / splits;
  const float inv_keep = 1.f / (1.f - drop_p);
  for (int n0 = 0; n0 < N; n0 += WG_N) {
    hipLaunchKernelGGL(wgrad_partial_kernel, dim3((K + 511) / 512, splits), dim3(256), 0, st, (const bf16*)G, ldg,
                       (const bf16*)X, ldx, ws, M, N, K, n0, rps, thresh16(drop_p), drop_seed);
    VLB_LAUNCH_CHECK();
  }
  const int64_t nk = (int64_t)N * K;
  int blocks = (int)((nk + 255) / 256); if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, ws, dW, splits, nk, alpha * inv_keep, beta);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

extern "C" int vlb_lora_down(const void* x, int ldx, const void* A, void* t, int ldt, int M, int K, int R, float scale,
                             float drop_p, const uint32_t* seeds_host, void* stream) {
  VLB_REQUIRE(x && A && t, "lora_down: null operand");
  VLB_REQUIRE(M > 0 && K % 32 == 0 && R % 16 == 0 && R >= 16 && R <= 48 && ldx % 8 == 0 && ldt % 4 == 0,
              "lora_down: bad shape M=%d K=%d R=%d", M, K, R);
  VLB_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seeds_host), "lora_down: bad dropout arguments");
  Seeds s{};
  for (int g = 0; g < R / 16; ++g) s.s[g] = seeds_host ? seeds_host[g] : 0u;
  const float sc = scale / (1.f - drop_p);
  dim3 grid((M + 63) / 64);
  hipStream_t st = as_stream(stream);
  const uint32_t th = thresh16(drop_p);
  switch (R / 16) {
    case 1: hipLaunchKernelGGL(lora_down_kernel<1>, grid, dim3(256), 0, st, (const bf16*)x, ldx, (const bf16*)A, (bf16*)t, ldt, M, K, sc, th, s); break;
    case 2: hipLaunchKernelGGL(lora_down_kernel<2>, grid, dim3(256), 0, st, (const bf16*)x, ldx, (const bf16*)A, (bf16*)t, ldt, M, K, sc, th, s); break;
    default: hipLaunchKernelGGL(lora_down_kernel<3>, grid, dim3(256), 0, st, (const bf16*)x, ldx, (const bf16*)A, (bf16*)t, ldt, M, K, sc, th, s); break;
  }
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

extern "C" int vlb_lora_dx_masked(const void* u, int ldu, const void* At, int ldat, void* dx, int lddx, int M, int K, int R,
                                  float drop_p, const uint32_t* seeds_host, void* stream) {
  VLB_REQUIRE(u && At && dx, "lora_dx_masked: null operand");
  VLB_REQUIRE(M > 0 && K % 16 == 0 && R % 16 == 0 && R >= 16 && R <= 48 && ldu % 8 == 0 && lddx % 4 == 0 && ldat % 8 == 0 && ldat >= R,
              "lora_dx_masked: bad shape M=%d K=%d R=%d", M, K, R);
  VLB_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seeds_host), "lora_dx_masked: bad dropout arguments");
  Seeds s{};
  for (int g = 0; g < R / 16; ++g) s.s[g] = seeds_host ? seeds_host[g] : 0u;
  dim3 grid((K + 1023) / 1024, (M + 15) / 16);
  hipStream_t st = as_stream(stream);
  const uint32_t th = thresh16(drop_p);
  const float ik = 1.f / (1.f - drop_p);
  switch (R / 16) {
    case 1: hipLaunchKernelGGL(lora_dx_kernel<1>, grid, dim3(256), 0, st, (const bf16*)u, ldu, (const bf16*)At, ldat, (bf16*)dx, lddx, M, K, ik, th, s); break;
    case 2: hipLaunchKernelGGL(lora_dx_kernel<2>, grid, dim3(256), 0, st, (const bf16*)u, ldu, (const bf16*)At, ldat, (bf16*)dx, lddx, M, K, ik, th, s); break;
    default: hipLaunchKernelGGL(lora_dx_kernel<3>, grid, dim3(256), 0, st, (const bf16*)u, ldu, (const bf16*)At, ldat, (bf16*)dx, lddx, M, K, ik, th, s); break;
  }
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
