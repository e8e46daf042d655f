// HBM-bound row / element-wise kernels of the path (norms, RoPE, SwiGLU, vision ingest, connector
// stencils, splice, weight mask, optimiser).  All use 16-byte per-lane accesses on bf16 data and
// fp32 arithmetic; none of them is reshaped into a GEMM.
#include "common.hpp"

namespace {

constexpr int kMaxBlocks = 256 * 8;  // grid cap for grid-stride element-wise kernels

__device__ __forceinline__ void load8(const bf16* p, float (&v)[8]) {
  const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
}
__device__ __forceinline__ void store8(bf16* p, const float (&v)[8]) {
  bf16x8 t;
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = (bf16)v[i];
  *reinterpret_cast<bf16x8*>(p) = t;
}

// ------------------------------------------------------------------ RMSNorm (one wave per row)
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                          bf16* __restrict__ y, int rows, int dim, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16* xr = x + (int64_t)row * dim;
  float ss = 0.f;
  for (int c = lane * 8; c < dim; c += 512) {
    float v[8]; load8(xr + c, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += v[i] * v[i];
  }
  const float rstd = rsqrtf(wave_sum(ss) / dim + eps);
  bf16* yr = y + (int64_t)row * dim;
  for (int c = lane * 8; c < dim; c += 512) {
    float v[8], g[8]; load8(xr + c, v); load8(w + c, g);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = g[i] * (float)(bf16)(v[i] * rstd);  // HF: w * bf16(x*rstd)
    store8(yr + c, v);
  }
}

// dx = rstd*(w*dy - xhat*mean(w*dy*xhat)) (+ dx_in)
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                          const bf16* __restrict__ dy, const bf16* __restrict__ dx_in,
                                                          bf16* __restrict__ dx, int rows, int dim, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16* xr = x + (int64_t)row * dim;
  const bf16* gr = dy + (int64_t)row * dim;
  float ss = 0.f, dot = 0.f;
  for (int c = lane * 8; c < dim; c += 512) {
    float v[8], g[8], ww[8]; load8(xr + c, v); load8(gr + c, g); load8(w + c, ww);
#pragma unroll
    for (int i = 0; i < 8; ++i) { ss += v[i] * v[i]; dot += v[i] * g[i] * ww[i]; }
  }
  const float rstd = rsqrtf(wave_sum(ss) / dim + eps);
  const float coef = wave_sum(dot) * rstd * rstd * rstd / dim;   // mean(w dy xhat) * rstd / ... folded
  bf16* dr = dx + (int64_t)row * dim;
  for (int c = lane * 8; c < dim; c += 512) {
    float v[8], g[8], ww[8], o[8]; load8(xr + c, v); load8(gr + c, g); load8(w + c, ww);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = rstd * ww[i] * g[i] - coef * v[i];
    if (dx_in) {
      float r[8]; load8(dx_in + (int64_t)row * dim + c, r);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] += r[i];
    }
    store8(dr + c, o);
  }
}

__device__ __forceinline__ float act_f(float x, int act) {
  switch (act) {
    case VLB_ACT_QUICK_GELU: return quick_gelu_f(x);
    case VLB_ACT_GELU: return gelu_erf_f(x);
    case VLB_ACT_SILU: return silu_f(x);
    default: return x;
  }
}

// ------------------------------------------------------------------ register-cached row kernels
// Rows of up to NI*512 elements are read ONCE: every lane issues its NI 16-byte loads up front (they are
// all in flight together), the statistics and the output are computed from registers.  Same arithmetic,
// same per-lane accumulation order as the generic kernels above/below - results are bit-identical.
template <int NI>
__global__ __launch_bounds__(256) void rmsnorm_fwd_cached_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                                 bf16* __restrict__ y, int rows, int dim, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16* xr = x + (int64_t)row * dim;
  bf16x8 xb[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) { const int c = lane * 8 + k * 512; xb[k] = c < dim ? *reinterpret_cast<const bf16x8*>(xr + c) : bf16x8{}; }
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < NI; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float v = (float)xb[k][i]; ss += v * v; }
  const float rstd = rsqrtf(wave_sum(ss) / dim + eps);
  bf16* yr = y + (int64_t)row * dim;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int c = lane * 8 + k * 512;
    if (c < dim) {
      float g[8], v[8]; load8(w + c, g);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = g[i] * (float)(bf16)((float)xb[k][i] * rstd);
      store8(yr + c, v);
    }
  }
}

// RMSNorm that also emits the MX-fp8 quantisation of its output (what the fp8 linear behind it consumes): y as above, plus
// q / s bit-identical to vlb_quantize_mxfp8(y) - the quantiser sees the bf16-rounded y.  dim % 32 == 0, dim <= NI*512.
template <int NI>
__global__ __launch_bounds__(256) void rmsnorm_fwd_q8_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w, bf16* __restrict__ y,
                                                            uint8_t* __restrict__ q, int ldq, uint8_t* __restrict__ s, int lds_,
                                                            int rows, int dim, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16* xr = x + (int64_t)row * dim;
  bf16x8 xb[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) { const int c = lane * 8 + k * 512; xb[k] = c < dim ? *reinterpret_cast<const bf16x8*>(xr + c) : bf16x8{}; }
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < NI; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float v = (float)xb[k][i]; ss += v * v; }
  const float rstd = rsqrtf(wave_sum(ss) / dim + eps);
  bf16* yr = y + (int64_t)row * dim;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int c = lane * 8 + k * 512;
    if (c < dim) {                       // (dim % 32 == 0: the four lanes of a block are in or out together)
      float g[8], v[8]; load8(w + c, g);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (float)(bf16)(g[i] * (float)(bf16)((float)xb[k][i] * rstd));
      store8(yr + c, v);
      mx_quantize_lane8(v, q + (int64_t)row * ldq, s + (int64_t)row * lds_, c, lane);
    }
  }
}

template <int NI>
__global__ __launch_bounds__(256) void rmsnorm_bwd_cached_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                                 const bf16* __restrict__ dy, const bf16* __restrict__ dx_in,
                                                                 bf16* __restrict__ dx, int rows, int dim, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16* xr = x + (int64_t)row * dim;
  const bf16* gr = dy + (int64_t)row * dim;
  bf16x8 xb[NI], gb[NI], rb[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int c = lane * 8 + k * 512;
    const bool in = c < dim;
    xb[k] = in ? *reinterpret_cast<const bf16x8*>(xr + c) : bf16x8{};
    gb[k] = in ? *reinterpret_cast<const bf16x8*>(gr + c) : bf16x8{};
    rb[k] = (in && dx_in) ? *reinterpret_cast<const bf16x8*>(dx_in + (int64_t)row * dim + c) : bf16x8{};
  }
  float ss = 0.f, dot = 0.f;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int c = lane * 8 + k * 512;
    if (c < dim) {
      float ww[8]; load8(w + c, ww);
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float v = (float)xb[k][i]; ss += v * v; dot += v * (float)gb[k][i] * ww[i]; }
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / dim + eps);
  const float coef = wave_sum(dot) * rstd * rstd * rstd / dim;
  bf16* dr = dx + (int64_t)row * dim;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int c = lane * 8 + k * 512;
    if (c < dim) {
      float ww[8], o[8]; load8(w + c, ww);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = rstd * ww[i] * (float)gb[k][i] - coef * (float)xb[k][i];
      if (dx_in) {
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] += (float)rb[k][i];
      }
      store8(dr + c, o);
    }
  }
}

template <int NI>
__global__ __launch_bounds__(256) void layernorm_fwd_cached_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                                   const bf16* __restrict__ b, const bf16* __restrict__ res,
                                                                   bf16* __restrict__ y, int rows, int dim, float eps, int act) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16* xr = x + (int64_t)row * dim;
  bf16x8 xb[NI], rb[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int c = lane * 8 + k * 512;
    xb[k] = c < dim ? *reinterpret_cast<const bf16x8*>(xr + c) : bf16x8{};
    rb[k] = (c < dim && res) ? *reinterpret_cast<const bf16x8*>(res + (int64_t)row * dim + c) : bf16x8{};
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NI; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (float)xb[k][i];
  const float mean = wave_sum(s) / dim;
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    if (lane * 8 + k * 512 < dim) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = (float)xb[k][i] - mean; ss += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / dim + eps);
  bf16* yr = y + (int64_t)row * dim;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int c = lane * 8 + k * 512;
    if (c < dim) {
      float v[8], g[8], bb[8]; load8(w + c, g); load8(b + c, bb);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = ((float)xb[k][i] - mean) * rstd * g[i] + bb[i];
      if (res) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += (float)rb[k][i];
      }
      if (act != VLB_ACT_NONE) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = act_f(v[i], act);
      }
      store8(yr + c, v);
    }
  }
}

// ------------------------------------------------------------------ LayerNorm (+residual, +act)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                            const bf16* __restrict__ b, const bf16* __restrict__ res,
                                                            bf16* __restrict__ y, int rows, int dim, float eps, int act) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16* xr = x + (int64_t)row * dim;
  float s = 0.f;
  for (int c = lane * 8; c < dim; c += 512) {
    float v[8]; load8(xr + c, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
  }
  const float mean = wave_sum(s) / dim;
  float ss = 0.f;
  for (int c = lane * 8; c < dim; c += 512) {
    float v[8]; load8(xr + c, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float d = v[i] - mean; ss += d * d; }
  }
  const float rstd = rsqrtf(wave_sum(ss) / dim + eps);
  bf16* yr = y + (int64_t)row * dim;
  for (int c = lane * 8; c < dim; c += 512) {
    float v[8], g[8], bb[8]; load8(xr + c, v); load8(w + c, g); load8(b + c, bb);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (v[i] - mean) * rstd * g[i] + bb[i];
    if (res) {
      float r[8]; load8(res + (int64_t)row * dim + c, r);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += r[i];
    }
    if (act != VLB_ACT_NONE) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = act_f(v[i], act);
    }
    store8(yr + c, v);
  }
}

// ------------------------------------------------------------------ RoPE (half rotation, in place)
__global__ void rope_kernel(bf16* __restrict__ x, int ld, const float* __restrict__ cs, const float* __restrict__ sn,
                            int S, int heads, int D, float sign, const int* __restrict__ pos, int64_t total) {
  const int half = D >> 1, cpr = half >> 3;  // 8-element chunks per half head
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int ch = idx % cpr;
    const int h = (idx / cpr) % heads;
    const int64_t tok = idx / ((int64_t)cpr * heads);   // row
    const int s = pos ? pos[tok] : (int)(tok % S);      // packed rows carry their position explicitly
    bf16* p = x + tok * ld + h * D + ch * 8;
    float a[8], b[8]; load8(p, a); load8(p + half, b);
    const float* c = cs + (int64_t)s * half + ch * 8;
    const float* n = sn + (int64_t)s * half + ch * 8;
    float oa[8], ob[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float co = c[i], si = sign * n[i];
      oa[i] = a[i] * co - b[i] * si;
      ob[i] = b[i] * co + a[i] * si;
    }
    store8(p, oa); store8(p + half, ob);
  }
}

// ------------------------------------------------------------------ SwiGLU
__global__ void swiglu_fwd_kernel(const bf16* __restrict__ gu, bf16* __restrict__ out, int ff, int64_t total) {
  const int cpr = ff >> 3;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / cpr; const int c = (idx % cpr) * 8;
    float g[8], u[8]; load8(gu + r * 2 * ff + c, g); load8(gu + r * 2 * ff + ff + c, u);
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = silu_f(g[i]) * u[i];
    store8(out + r * ff + c, g);
  }
}
// SwiGLU forward that also emits the MX-fp8 quantisation of its output (the down projection's operand); ff % 32 == 0, so
// four consecutive work items (= lanes) always cover one 32-element block of one row.
__global__ void swiglu_fwd_q8_kernel(const bf16* __restrict__ gu, bf16* __restrict__ out, uint8_t* __restrict__ q, int ldq,
                                     uint8_t* __restrict__ s, int lds_, int ff, int64_t total) {
  const int cpr = ff >> 3;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / cpr; const int c = (idx % cpr) * 8;
    float g[8], u[8]; load8(gu + r * 2 * ff + c, g); load8(gu + r * 2 * ff + ff + c, u);
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = (float)(bf16)(silu_f(g[i]) * u[i]);
    store8(out + r * ff + c, g);
    mx_quantize_lane8(g, q + r * ldq, s + r * lds_, c, threadIdx.x & 63);
  }
}
__global__ void swiglu_bwd_kernel(const bf16* __restrict__ gu, const bf16* __restrict__ dout, bf16* __restrict__ dgu,
                                  int ff, int64_t total) {
  const int cpr = ff >> 3;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / cpr; const int c = (idx % cpr) * 8;
    float g[8], u[8], d[8], dg[8], du[8];
    load8(gu + r * 2 * ff + c, g); load8(gu + r * 2 * ff + ff + c, u); load8(dout + r * ff + c, d);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float sg = sigmoid_f(g[i]);
      const float si = g[i] * sg;
      du[i] = d[i] * si;
      dg[i] = d[i] * u[i] * (sg * (1.f + g[i] * (1.f - sg)));
    }
    store8(dgu + r * 2 * ff + c, dg); store8(dgu + r * 2 * ff + ff + c, du);
  }
}
__global__ void add_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, bf16* __restrict__ y, int64_t n8) {
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < n8; idx += (int64_t)gridDim.x * blockDim.x) {
    float x[8], z[8]; load8(a + idx * 8, x); load8(b + idx * 8, z);
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] += z[i];
    store8(y + idx * 8, x);
  }
}

// ------------------------------------------------------------------ vision ingest
// one thread per 8 output columns of a patch row; column k = c*P*P + py*P + px
__global__ void patchify_kernel(const float* __restrict__ vis, bf16* __restrict__ out, int H, int W, int P, int Kpad,
                                int64_t total) {
  const int G = W / P, GH = H / P, K = 3 * P * P, cpr = Kpad >> 3;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int kc = (idx % cpr) * 8;
    const int64_t prow = idx / cpr;
    const int gx = prow % G, gy = (prow / G) % GH;
    const int64_t n = prow / ((int64_t)G * GH);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = kc + i;
      if (k < K) {
        const int c = k / (P * P), py = (k / P) % P, px = k % P;
        v[i] = vis[((n * 3 + c) * H + gy * P + py) * (int64_t)W + gx * P + px];
      } else {
        v[i] = 0.f;
      }
    }
    store8(out + prow * Kpad + kc, v);
  }
}
__global__ void vit_assemble_kernel(const bf16* __restrict__ pe, const bf16* __restrict__ cls, const bf16* __restrict__ pos,
                                    bf16* __restrict__ tok, int G, int D, int64_t total) {
  const int cpr = D >> 3;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (idx % cpr) * 8;
    const int t = (idx / cpr) % (G + 1);
    const int64_t n = idx / ((int64_t)cpr * (G + 1));
    float a[8], p[8];
    if (t == 0) load8(cls + c, a); else load8(pe + (n * G + t - 1) * D + c, a);
    load8(pos + (int64_t)t * D + c, p);
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] += p[i];
    store8(tok + (n * (G + 1) + t) * D + c, a);
  }
}
__global__ void drop_cls_kernel(const bf16* __restrict__ tok, bf16* __restrict__ out, int G, int D, int64_t total) {
  const int cpr = D >> 3;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (idx % cpr) * 8;
    const int t = (idx / cpr) % G;
    const int64_t n = idx / ((int64_t)cpr * G);
    *reinterpret_cast<bf16x8*>(out + (n * G + t) * D + c) =
        *reinterpret_cast<const bf16x8*>(tok + (n * (G + 1) + t + 1) * D + c);
  }
}

// ------------------------------------------------------------------ connector stencils (NHWC)
// LDS-tiled: one block = one image x 32 channels.  The whole H x W plane of those channels (64 bytes per
// position) is staged once in LDS and every output reads its 9 taps from there, so HBM sees each input
// byte once instead of nine L2 re-reads.  grid (C/32, N), block 256; LDS = H*W*64 bytes.
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w9,
                                                        bf16* __restrict__ y, int H, int W, int C) {
  extern __shared__ __attribute__((aligned(16))) char plane[];   // [H*W][4 chunks][16 B]
  const int c0 = blockIdx.x * 32;
  const int64_t n = blockIdx.y;
  const int HW = H * W;
  const bf16* xb = x + n * HW * (int64_t)C + c0;
  for (int i = threadIdx.x; i < HW * 4; i += 256) {
    const int p = i >> 2, ch = i & 3;
    *reinterpret_cast<bf16x8*>(plane + i * 16) = *reinterpret_cast<const bf16x8*>(xb + (int64_t)p * C + ch * 8);
  }
  const int ch = threadIdx.x & 3;
  float k[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t) load8(w9 + (int64_t)t * C + c0 + ch * 8, k[t]);
  __syncthreads();
  bf16* yb = y + n * HW * (int64_t)C + c0 + ch * 8;
  for (int p = threadIdx.x >> 2; p < HW; p += 64) {
    const int hy = p / W, wx = p % W;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int yy = hy + dy, xx = wx + dx;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        float v[8]; load8(reinterpret_cast<const bf16*>(plane + ((yy * W + xx) * 4 + ch) * 16), v);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += v[i] * k[(dy + 1) * 3 + (dx + 1)][i];
      }
    store8(yb + (int64_t)p * C, acc);
  }
}
// block = 64 channel chunks x 4 row groups; grid (C/512, N)
__global__ __launch_bounds__(256) void se_pool_kernel(const bf16* __restrict__ x, bf16* __restrict__ pooled, int HW, int C) {
  __shared__ float red[4][64][8];
  const int cc = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cc) * 8;
  const int64_t n = blockIdx.y;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c < C) {
    for (int r = rg; r < HW; r += 4) {
      float v[8]; load8(x + (n * HW + r) * C + c, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += v[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[rg][cc][i] = acc[i];
  __syncthreads();
  if (rg == 0 && c < C) {
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (red[0][cc][i] + red[1][cc][i] + red[2][cc][i] + red[3][cc][i]) / HW;
    store8(pooled + n * C + c, o);
  }
}
__global__ void se_scale_kernel(const bf16* __restrict__ x, const bf16* __restrict__ gate, bf16* __restrict__ y, int HW,
                                int C, int64_t total) {
  const int cpr = C >> 3;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (idx % cpr) * 8;
    const int64_t n = idx / ((int64_t)cpr * HW);
    float v[8], g[8]; load8(x + idx * 8, v); load8(gate + n * C + c, g);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= sigmoid_f(g[i]);
    store8(y + idx * 8, v);
  }
}
__global__ void im2col3d_kernel(const bf16* __restrict__ x, bf16* __restrict__ cols, int T, int H, int W, int C, int T2,
                                int H2, int W2, int64_t total) {
  const int cpr = C >> 3;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (idx % cpr) * 8;
    const int tap = (idx / cpr) % 8;
    const int64_t orow = idx / ((int64_t)cpr * 8);
    const int w2 = orow % W2, h2 = (orow / W2) % H2, t2 = (orow / ((int64_t)W2 * H2)) % T2;
    const int64_t b = orow / ((int64_t)W2 * H2 * T2);
    const int t = 2 * t2 - 1 + (tap >> 2), h = 2 * h2 - 1 + ((tap >> 1) & 1), w = 2 * w2 - 1 + (tap & 1);
    bf16x8 v = {};
    if (t >= 0 && t < T && h >= 0 && h < H && w >= 0 && w < W)
      v = *reinterpret_cast<const bf16x8*>(x + (((b * T + t) * H + h) * W + w) * C + c);
    *reinterpret_cast<bf16x8*>(cols + (orow * 8 + tap) * C + c) = v;
  }
}

// ------------------------------------------------------------------ splice + weight mask
// grid (ceil(S/16), B), block 256: each block locates the video slot of its row, then copies 16 rows.
__global__ __launch_bounds__(256) void splice_kernel(const int64_t* __restrict__ ids, const bf16* __restrict__ emb,
                                                     const bf16* __restrict__ vid, bf16* __restrict__ out,
                                                     uint8_t* __restrict__ kmask, int* __restrict__ err, int L, int Nv,
                                                     int D, int64_t video_id, int vocab, const int* __restrict__ cu,
                                                     int* __restrict__ rowpos) {
  __shared__ int s_pos, s_cnt;
  const int b = blockIdx.y, Sfull = L - 1 + Nv;
  // packed layout: clip b owns rows [cu[b], cu[b+1]) = its first (unpadded) tokens; dense layout: b*Sfull
  const int S = cu ? cu[b + 1] - cu[b] : Sfull;
  const int64_t row0 = cu ? cu[b] : (int64_t)b * Sfull;
  if (threadIdx.x == 0) { s_pos = L; s_cnt = 0; }
  __syncthreads();
  for (int i = threadIdx.x; i < L; i += blockDim.x)
    if (ids[(int64_t)b * L + i] == video_id) { atomicMin(&s_pos, i); atomicAdd(&s_cnt, 1); }
  __syncthreads();
  const int pos = s_pos;
  if (s_cnt != 1) {
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicExch(err, 1);
    if (s_cnt == 0) return;
  }
  const int cpr = D >> 3;
  const int s0 = blockIdx.x * 16;
  for (int i = threadIdx.x; i < 16 * cpr; i += blockDim.x) {
    const int s = s0 + i / cpr, c = (i % cpr) * 8;
    if (s >= S) break;
    const bf16* src;
    if (s >= pos && s < pos + Nv) {
      src = vid + ((int64_t)b * Nv + (s - pos)) * D;
    } else {
      int64_t id = ids[(int64_t)b * L + (s < pos ? s : s - Nv + 1)];
      if (id < 0 || id >= vocab) { atomicExch(err, 2); id = 0; }
      src = emb + id * D;
    }
    *reinterpret_cast<bf16x8*>(out + (row0 + s) * D + c) = *reinterpret_cast<const bf16x8*>(src + c);
  }
  for (int i = threadIdx.x; i < 16; i += blockDim.x) {
    const int s = s0 + i;
    if (s < S) {
      kmask[row0 + s] = (s < Nv - 1) ? 1 : (ids[(int64_t)b * L + s - (Nv - 1)] != 0);
      if (rowpos) rowpos[row0 + s] = s;
    }
  }
}

__global__ void weight_mask_kernel(const int64_t* __restrict__ pv, const double* __restrict__ vw,
                                   const double* __restrict__ lw, float* __restrict__ out, int F, int Lw, int tpf, int S,
                                   int round_bf16, int64_t total) {
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int s = idx % S; const int64_t b = idx / S;
    const int pad = (int)pv[b * 3], inst = (int)pv[b * 3 + 1], dia = (int)pv[b * 3 + 2];
    const int nvis = F * tpf;
    const int tail = nvis + 2 + inst + dia + 4 + pad;
    const int t = s - (S - tail);
    float v = 0.f;
    if (t >= 0) {
      if (t < nvis) v = (float)vw[b * F + t / tpf];
      else {
        const int u = t - nvis - 2 - inst;
        if (u >= 0 && u < dia && u < Lw) v = (float)lw[b * Lw + u];
      }
    }
    out[idx] = round_bf16 ? (float)(bf16)v : v;   // the reference builds the mask in bf16 (litmodule :190-194)
  }
}

// ------------------------------------------------------------------ casts / optimiser
__global__ void cast_f2b_kernel(const float* __restrict__ in, bf16* __restrict__ out, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (bf16)in[i];
}
__global__ void cast_b2f_kernel(const bf16* __restrict__ in, float* __restrict__ out, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (float)in[i];
}
// two-stage, fixed-order reduction: every rank of a data-parallel job computes bit-identical norms
constexpr int kSumsqBlocks = 1024;
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
  __shared__ float red[16];
  float s = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += g[i] * g[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] += s;
}
__global__ void adamw_kernel(float* __restrict__ p, bf16* __restrict__ pb, const float* __restrict__ g,
                             float* __restrict__ m, float* __restrict__ v, int64_t n, float lr, float b1, float b2,
                             float eps, float wd, float bc1, float bc2_sqrt, const float* __restrict__ sumsq, float max_norm) {
  float clip = 1.f;
  if (max_norm > 0.f && sumsq) clip = fminf(1.f, max_norm / (sqrtf(sumsq[0]) + 1e-6f));
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * clip;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (pb) pb[i] = (bf16)pi;
  }
}

inline int grid_for(int64_t n, int block) {
  int64_t b = (n + block - 1) / block;
  return (int)(b < 1 ? 1 : (b > kMaxBlocks ? kMaxBlocks : b));
}
}  // namespace

#define ROWS_KERNEL_CHECK(name)                                                            \
  VLB_REQUIRE(rows > 0 && dim > 0 && dim % 8 == 0, name ": rows=%d dim=%d (dim must be a multiple of 8)", rows, dim)

extern "C" int vlb_rmsnorm_fwd(const void* x, const void* w, void* y, int rows, int dim, float eps, void* stream) {
  ROWS_KERNEL_CHECK("rmsnorm_fwd");
#define VLB_RN_ARGS dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), (const bf16*)x, (const bf16*)w, (bf16*)y, rows, dim, eps
  if (dim <= 512) hipLaunchKernelGGL(rmsnorm_fwd_cached_kernel<1>, VLB_RN_ARGS);
  else if (dim <= 4096) hipLaunchKernelGGL(rmsnorm_fwd_cached_kernel<8>, VLB_RN_ARGS);
  else hipLaunchKernelGGL(rmsnorm_fwd_kernel, VLB_RN_ARGS);
#undef VLB_RN_ARGS
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_rmsnorm_fwd_mxfp8(const void* x, const void* w, void* y, void* q, int ldq, void* scales, int lds, int rows, int dim,
                                     float eps, void* stream) {
  ROWS_KERNEL_CHECK("rmsnorm_fwd_mxfp8");
  VLB_REQUIRE(q && scales && dim % 32 == 0 && dim <= 4096 && ldq >= dim && ldq % 8 == 0 && lds >= dim / 32 && ((uintptr_t)q % 8) == 0,
              "rmsnorm_fwd_mxfp8: dim must be a multiple of 32, at most 4096; q rows 8-byte aligned (dim=%d ldq=%d)", dim, ldq);
#define VLB_RQ_ARGS dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), (const bf16*)x, (const bf16*)w, (bf16*)y, (uint8_t*)q, ldq, \
                    (uint8_t*)scales, lds, rows, dim, eps
  if (dim <= 512) hipLaunchKernelGGL(rmsnorm_fwd_q8_kernel<1>, VLB_RQ_ARGS);
  else hipLaunchKernelGGL(rmsnorm_fwd_q8_kernel<8>, VLB_RQ_ARGS);
#undef VLB_RQ_ARGS
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_rmsnorm_bwd(const void* x, const void* w, const void* dy, const void* dx_in, void* dx, int rows,
                               int dim, float eps, void* stream) {
  ROWS_KERNEL_CHECK("rmsnorm_bwd");
#define VLB_RB_ARGS dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), (const bf16*)x, (const bf16*)w, (const bf16*)dy, \
                    (const bf16*)dx_in, (bf16*)dx, rows, dim, eps
  // the register-cached form loses here (three cached operands per row halve the occupancy: 52.8 vs 37.0 us
  // at [10240x4096]); it is kept for short rows only
  if (dim <= 512) hipLaunchKernelGGL(rmsnorm_bwd_cached_kernel<1>, VLB_RB_ARGS);
  else hipLaunchKernelGGL(rmsnorm_bwd_kernel, VLB_RB_ARGS);
#undef VLB_RB_ARGS
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_layernorm_fwd(const void* x, const void* w, const void* b, const void* residual, void* y, int rows,
                                 int dim, float eps, int act, void* stream) {
  ROWS_KERNEL_CHECK("layernorm_fwd");
#define VLB_LN_ARGS dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), (const bf16*)x, (const bf16*)w, (const bf16*)b, \
                    (const bf16*)residual, (bf16*)y, rows, dim, eps, act
  if (dim <= 512) hipLaunchKernelGGL(layernorm_fwd_cached_kernel<1>, VLB_LN_ARGS);
  else if (dim <= 1024) hipLaunchKernelGGL(layernorm_fwd_cached_kernel<2>, VLB_LN_ARGS);
  else if (dim <= 4096) hipLaunchKernelGGL(layernorm_fwd_cached_kernel<8>, VLB_LN_ARGS);
  else hipLaunchKernelGGL(layernorm_fwd_kernel, VLB_LN_ARGS);
#undef VLB_LN_ARGS
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_rope_inplace(void* x, int ld, const float* cos_t, const float* sin_t, int rows, int S, int heads, int D,
                                int sign, const int* pos, void* stream) {
  VLB_REQUIRE(D % 16 == 0 && ld % 8 == 0 && rows > 0 && S > 0 && heads > 0, "rope: bad shape D=%d ld=%d", D, ld);
  const int64_t total = (int64_t)rows * heads * (D / 16);
  hipLaunchKernelGGL(rope_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), (bf16*)x, ld, cos_t,
                     sin_t, S, heads, D, sign >= 0 ? 1.f : -1.f, pos, total);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_swiglu_fwd(const void* gu, void* out, int rows, int ff, void* stream) {
  VLB_REQUIRE(rows > 0 && ff % 8 == 0, "swiglu: ff=%d must be a multiple of 8", ff);
  const int64_t total = (int64_t)rows * (ff / 8);
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), (const bf16*)gu,
                     (bf16*)out, ff, total);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_swiglu_fwd_mxfp8(const void* gu, void* out, void* q, int ldq, void* scales, int lds, int rows, int ff, void* stream) {
  VLB_REQUIRE(rows > 0 && ff % 32 == 0 && q && scales && ldq >= ff && ldq % 8 == 0 && lds >= ff / 32 && ((uintptr_t)q % 8) == 0,
              "swiglu_fwd_mxfp8: ff=%d must be a multiple of 32; q rows 8-byte aligned", ff);
  const int64_t total = (int64_t)rows * (ff / 8);
  hipLaunchKernelGGL(swiglu_fwd_q8_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), (const bf16*)gu, (bf16*)out,
                     (uint8_t*)q, ldq, (uint8_t*)scales, lds, ff, total);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_swiglu_bwd(const void* gu, const void* dout, void* dgu, int rows, int ff, void* stream) {
  VLB_REQUIRE(rows > 0 && ff % 8 == 0, "swiglu_bwd: ff=%d must be a multiple of 8", ff);
  const int64_t total = (int64_t)rows * (ff / 8);
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), (const bf16*)gu,
                     (const bf16*)dout, (bf16*)dgu, ff, total);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_add_bf16(const void* a, const void* b, void* y, int64_t n, void* stream) {
  VLB_REQUIRE(n > 0 && n % 8 == 0, "add: n must be a positive multiple of 8");
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, as_stream(stream), (const bf16*)a,
                     (const bf16*)b, (bf16*)y, n / 8);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_patchify(const float* vision, void* patches, int N, int H, int W, int P, int Kpad, void* stream) {
  VLB_REQUIRE(N > 0 && P > 0 && H % P == 0 && W % P == 0, "patchify: H,W must be multiples of P");
  VLB_REQUIRE(Kpad % 8 == 0 && Kpad >= 3 * P * P, "patchify: Kpad=%d must be a multiple of 8 and >= 3*P*P", Kpad);
  const int64_t total = (int64_t)N * (H / P) * (W / P) * (Kpad / 8);
  hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), vision,
                     (bf16*)patches, H, W, P, Kpad, total);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_vit_assemble(const void* patch_emb, const void* cls, const void* pos, void* tokens, int N, int G,
                                int D, void* stream) {
  VLB_REQUIRE(N > 0 && G > 0 && D % 8 == 0, "vit_assemble: bad shape");
  const int64_t total = (int64_t)N * (G + 1) * (D / 8);
  hipLaunchKernelGGL(vit_assemble_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream),
                     (const bf16*)patch_emb, (const bf16*)cls, (const bf16*)pos, (bf16*)tokens, G, D, total);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_drop_cls(const void* tokens, void* out, int N, int G, int D, void* stream) {
  VLB_REQUIRE(N > 0 && G > 0 && D % 8 == 0, "drop_cls: bad shape");
  const int64_t total = (int64_t)N * G * (D / 8);
  hipLaunchKernelGGL(drop_cls_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), (const bf16*)tokens,
                     (bf16*)out, G, D, total);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_dwconv3x3(const void* x, const void* w, void* y, int N, int H, int W, int C, void* stream) {
  VLB_REQUIRE(N > 0 && H > 0 && W > 0 && C % 32 == 0, "dwconv3x3: bad shape (C must be a multiple of 32)");
  const int lds = H * W * 64;
  VLB_REQUIRE(lds <= 160 * 1024, "dwconv3x3: plane of %dx%d positions does not fit in LDS", H, W);
  static int reserved = 0;
  if (lds > reserved) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { vlb_set_error("dwconv3x3: LDS reservation failed: %s", hipGetErrorString(e)); return VLB_ERR_LAUNCH; }
    reserved = lds;
  }
  hipLaunchKernelGGL(dwconv3x3_kernel, dim3(C / 32, N), dim3(256), lds, as_stream(stream), (const bf16*)x, (const bf16*)w,
                     (bf16*)y, H, W, C);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_se_pool(const void* x, void* pooled, int N, int HW, int C, void* stream) {
  VLB_REQUIRE(N > 0 && HW > 0 && C % 8 == 0, "se_pool: bad shape");
  hipLaunchKernelGGL(se_pool_kernel, dim3((C / 8 + 63) / 64, N), dim3(256), 0, as_stream(stream), (const bf16*)x,
                     (bf16*)pooled, HW, C);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_se_scale(const void* x, const void* gate, void* y, int N, int HW, int C, void* stream) {
  VLB_REQUIRE(N > 0 && HW > 0 && C % 8 == 0, "se_scale: bad shape");
  const int64_t total = (int64_t)N * HW * (C / 8);
  hipLaunchKernelGGL(se_scale_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), (const bf16*)x,
                     (const bf16*)gate, (bf16*)y, HW, C, total);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_im2col3d_k2s2p1(const void* x, void* cols, int B, int T, int H, int W, int C, void* stream) {
  VLB_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0 && C % 8 == 0, "im2col3d: bad shape");
  const int T2 = T / 2 + 1, H2 = H / 2 + 1, W2 = W / 2 + 1;
  const int64_t total = (int64_t)B * T2 * H2 * W2 * 8 * (C / 8);
  hipLaunchKernelGGL(im2col3d_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), (const bf16*)x,
                     (bf16*)cols, T, H, W, C, T2, H2, W2, total);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_splice_embed(const int64_t* ids, const void* embed_w, const void* video_tokens, void* embeds,
                                uint8_t* key_mask, int* err_flag, int B, int L, int Nv, int D, int64_t video_id,
                                int vocab, const int* cu_rows, int* row_pos, void* stream) {
  VLB_REQUIRE(B > 0 && L > 0 && Nv > 0 && D % 8 == 0 && err_flag, "splice: bad shape");
  const int S = L - 1 + Nv;
  hipLaunchKernelGGL(splice_kernel, dim3((S + 15) / 16, B), dim3(256), 0, as_stream(stream), ids,
                     (const bf16*)embed_w, (const bf16*)video_tokens, (bf16*)embeds, key_mask, err_flag, L, Nv, D,
                     video_id, vocab, cu_rows, row_pos);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_weight_mask(const int64_t* padvals, const double* vis_w, const double* lang_w, float* wmask, int B,
                               int F, int Lw, int tokens_per_frame, int S, int round_bf16, void* stream) {
  VLB_REQUIRE(B > 0 && F > 0 && Lw >= 0 && tokens_per_frame > 0 && S > 0, "weight_mask: bad shape");
  const int64_t total = (int64_t)B * S;
  hipLaunchKernelGGL(weight_mask_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), padvals, vis_w,
                     lang_w, wmask, F, Lw, tokens_per_frame, S, round_bf16, total);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream) {
  VLB_REQUIRE(n > 0, "cast: n must be positive");
  hipLaunchKernelGGL(cast_f2b_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), in, (bf16*)out, n);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_cast_bf16_to_f32(const void* in, float* out, int64_t n, void* stream) {
  VLB_REQUIRE(n > 0, "cast: n must be positive");
  hipLaunchKernelGGL(cast_b2f_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), (const bf16*)in, out, n);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
// Head dropout (litmodule :226,251): keep/(1-p) per element from a counter-based hash of (seed, index) - the
// same lowbias32 mixer as the LoRA dropout masks (lora.hip), 16 random bits per element, so the mask is a pure
// function of (seed, position): nothing to store, and a resumed run needs only the step counter behind the seed.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__global__ void dropout_keep_scale_kernel(float* __restrict__ out, int64_t n, uint32_t key, uint32_t thresh, float inv_keep) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i >= n) return;
  const uint32_t h = mix32((uint32_t)(i >> 1) ^ key);
  out[i] = (h & 0xffffu) >= thresh ? inv_keep : 0.f;
  if (i + 1 < n) out[i + 1] = (h >> 16) >= thresh ? inv_keep : 0.f;
}
static inline uint32_t mix32_host(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
extern "C" int vlb_dropout_keep_scale(float* out, int64_t n, float p, uint32_t seed, void* stream) {
  VLB_REQUIRE(out && n > 0 && n < ((int64_t)1 << 32) && p >= 0.f && p < 1.f, "dropout_keep_scale: bad args");
  const uint32_t thresh = (uint32_t)(p * 65536.f + 0.5f);
  hipLaunchKernelGGL(dropout_keep_scale_kernel, dim3(grid_for((n + 1) / 2, 256)), dim3(256), 0, as_stream(stream), out, n,
                     mix32_host(seed), thresh, 1.f / (1.f - p));
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_sumsq_ws_floats(void) { return kSumsqBlocks; }
extern "C" int vlb_grad_sumsq(const float* g, int64_t n, float* sumsq, float* ws, void* stream) {
  VLB_REQUIRE(n > 0 && g && sumsq && ws, "grad_sumsq: bad args");
  hipStream_t st = as_stream(stream);
  int64_t nb64 = (n + 1023) / 1024;
  const int nb = (int)(nb64 > kSumsqBlocks ? kSumsqBlocks : nb64);
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, st, g, n, ws);
  VLB_LAUNCH_CHECK();
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, ws, nb, sumsq);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_adamw_step(float* master, void* param_bf16, const float* grad, float* m, float* v, int64_t n,
                              float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                              const float* sumsq, float max_norm, void* stream) {
  VLB_REQUIRE(n > 0 && master && grad && m && v && step >= 1, "adamw: bad args");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), master, (bf16*)param_bf16,
                     grad, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, sumsq, max_norm);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
