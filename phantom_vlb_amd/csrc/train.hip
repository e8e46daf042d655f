// Full-parameter fine-tuning pieces (BASELINE configs[4]; reference litmodule :86-99: freeze_backbone=False,
// use_lora=False trains everything but the vision tower).  Everything here is HBM-bound byte work around the MFMA
// GEMMs: the weight gradients themselves are the same TN GEMM `dW[N,K] = dy^T[N,M] . x^T[K,M]^T` on transposed
// activations (vlb_transpose_pad lays them down, token axis padded with zeros to the GEMM's K granule).
//
//   vlb_transpose_pad      out[C, Rpad] = in[R, C]^T, columns R..Rpad-1 zero
//   vlb_rmsnorm_bwd_dw     d gamma of RMSNorm: sum_rows dy * x * rstd (fixed-order two-stage reduction)
//   vlb_embed_grad         d embed_tokens: per-token sum of d(inputs_embeds) rows, occurrence lists from the host
//   vlb_grad_sumsq_bf16 / vlb_adamw_step_g16   clip norm and AdamW reading bf16 gradients (what the wgrad GEMMs write)
//   vlb_colsum             bias gradients: column sums of a [rows, C] bf16 matrix
//   vlb_layernorm_bwd      LayerNorm (+ residual, + activation) backward: dx, d residual, d gamma / d beta partials
//   vlb_act_bwd            dx = dy * act'(x) for SiLU / GELU
//   vlb_dwconv3x3_bwd_w    depthwise 3x3 weight gradient
//   vlb_se_bwd             squeeze-excite: d gate logits and d input of x * sigmoid(s)
//   vlb_col2im3d_k2s2p1    inverse of the Conv3d im2col (every input element sits in exactly one window)
#include "common.hpp"

namespace {
__device__ __forceinline__ void ld8(const bf16* p, float (&v)[8]) {
  const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
}
__device__ __forceinline__ void st8(bf16* p, const float (&v)[8]) {
  bf16x8 t;
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = (bf16)v[i];
  *reinterpret_cast<bf16x8*>(p) = t;
}

// ---------------------------------------------------------------- transpose with zero padding of the token axis
// 64x64 tiles through LDS; 16-byte global loads and stores on both sides.
__global__ __launch_bounds__(256) void transpose_pad_kernel(const bf16* __restrict__ in, int ld_in, bf16* __restrict__ out, int ld_out,
                                                            int R, int C, int Rpad) {
  __shared__ bf16 tile[64][72];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int t = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int r = (t >> 3) + 32 * p, cc = (t & 7) * 8;
    bf16x8 v{};
    if (r0 + r < R) {
      if (c0 + cc + 8 <= C) v = *reinterpret_cast<const bf16x8*>(in + (int64_t)(r0 + r) * ld_in + c0 + cc);
      else for (int j = 0; j < 8; ++j) if (c0 + cc + j < C) v[j] = in[(int64_t)(r0 + r) * ld_in + c0 + cc + j];
    }
    *reinterpret_cast<bf16x8*>(&tile[r][cc]) = v;       // rows past R stay zero: the padded token columns
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int c = (t >> 3) + 32 * p, rr = (t & 7) * 8;
    if (c0 + c >= C || r0 + rr >= Rpad) continue;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = tile[rr + j][c];
    *reinterpret_cast<bf16x8*>(out + (int64_t)(c0 + c) * ld_out + r0 + rr) = v;
  }
}

// ---------------------------------------------------------------- RMSNorm d gamma
// One wave per row, RB rows per block; lane l owns columns 8l + 512k.  part[block][dim] fp32, then a fixed-order sum.
// With w != nullptr the same pass also writes dx = rstd * (w dy) - x * mean(w dy x) rstd^3 (+ dx_in): RMSNorm's whole
// backward in one sweep over x and dy (vlb_rmsnorm_bwd_full).
template <int NI>
__global__ __launch_bounds__(256) void rmsnorm_dw_partial_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                                                 float* __restrict__ part, int rows, int dim, float eps, int rows_per_block,
                                                                 const bf16* __restrict__ w = nullptr, const bf16* __restrict__ dx_in = nullptr,
                                                                 bf16* __restrict__ dx = nullptr) {
  extern __shared__ float red[];                      // [4][dim]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float acc[NI][8];
#pragma unroll
  for (int k = 0; k < NI; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[k][i] = 0.f;
  const int rbeg = blockIdx.x * rows_per_block, rend = min(rows, rbeg + rows_per_block);
  for (int row = rbeg + wave; row < rend; row += 4) {
    const bf16* xr = x + (int64_t)row * dim;
    const bf16* gr = dy + (int64_t)row * dim;
    float xv[NI][8], gv[NI][8];
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int c = lane * 8 + k * 512;
      if (c < dim) { ld8(xr + c, xv[k]); ld8(gr + c, gv[k]); }
      else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { xv[k][i] = 0.f; gv[k][i] = 0.f; }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) ss += xv[k][i] * xv[k][i];
    }
    const float rstd = rsqrtf(wave_sum(ss) / dim + eps);
#pragma unroll
    for (int k = 0; k < NI; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[k][i] += gv[k][i] * xv[k][i] * rstd;
    if (dx) {
      float wv[NI][8];
      float dot = 0.f;
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const int c = lane * 8 + k * 512;
        if (c < dim) ld8(w + c, wv[k]);
        else {
#pragma unroll
          for (int i = 0; i < 8; ++i) wv[k][i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) dot += xv[k][i] * gv[k][i] * wv[k][i];
      }
      const float coef = wave_sum(dot) * rstd * rstd * rstd / dim;
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const int c = lane * 8 + k * 512;
        if (c >= dim) continue;
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = rstd * wv[k][i] * gv[k][i] - coef * xv[k][i];
        if (dx_in) {
          float r[8]; ld8(dx_in + (int64_t)row * dim + c, r);
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] += r[i];
        }
        st8(dx + (int64_t)row * dim + c, o);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int c = lane * 8 + k * 512;
    if (c < dim)
#pragma unroll
      for (int i = 0; i < 8; ++i) red[wave * dim + c + i] = acc[k][i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < dim; c += 256)
    part[(int64_t)blockIdx.x * dim + c] = red[c] + red[dim + c] + red[2 * dim + c] + red[3 * dim + c];
}
// out[c] = sum_b part[b][c]: a block owns 32 columns; 8 row groups of threads each sum the partials b = g, g+8, ...
// (ascending), the 8 group sums are added in a fixed order: deterministic, and nb/8 dependent loads per thread
// instead of nb.
__global__ __launch_bounds__(256) void colpart_reduce_kernel(const float* __restrict__ part, int nb, int dim, bf16* __restrict__ out_bf16,
                                                             float* __restrict__ out_f32) {
  __shared__ float red[8][32];
  const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (c < dim)
    for (int b = grp; b < nb; b += 8) s += part[(int64_t)b * dim + c];
  red[grp][cl] = s;
  __syncthreads();
  if (grp == 0 && c < dim) {
    float t = red[0][cl];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][cl];
    if (out_bf16) out_bf16[c] = (bf16)t;
    if (out_f32) out_f32[c] = t;
  }
}

// ---------------------------------------------------------------- column sums (bias gradients) / two-column variants
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16* __restrict__ x, int ld, float* __restrict__ part, int rows, int dim,
                                                             int rows_per_block) {
  // thread t owns 8 columns c = 8t + 2048k; rows of the block are walked in order
  const int rbeg = blockIdx.y * rows_per_block, rend = min(rows, rbeg + rows_per_block);
  const int c = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (c >= dim) return;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r = rbeg; r < rend; ++r) {
    float v[8]; ld8(x + (int64_t)r * ld + c, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += v[i];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) part[(int64_t)blockIdx.y * dim + c + i] = acc[i];
}

// ---------------------------------------------------------------- embedding gradient
// job j: token tok[j] with occurrences rows[beg[j] .. beg[j+1]) of d_embeds (ascending: fixed summation order)
__global__ __launch_bounds__(256) void embed_grad_kernel(const bf16* __restrict__ dx, int ld, const int* __restrict__ tok,
                                                         const int* __restrict__ beg, const int* __restrict__ rows, bf16* __restrict__ dE,
                                                         int D) {
  const int j = blockIdx.x;
  const int b0 = beg[j], b1 = beg[j + 1];
  for (int c = threadIdx.x * 8; c < D; c += 256 * 8) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int o = b0; o < b1; ++o) {
      float v[8]; ld8(dx + (int64_t)rows[o] * ld + c, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += v[i];
    }
    st8(dE + (int64_t)tok[j] * D + c, acc);
  }
}

// ---------------------------------------------------------------- optimiser on bf16 gradients
constexpr int kSumsqBlocksB = 1024;
__global__ __launch_bounds__(256) void sumsq_bf16_kernel(const bf16* __restrict__ g, int64_t n8, float* __restrict__ part) {
  __shared__ float red[16];
  float s = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float v[8]; ld8(g + i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k] * v[k];
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sumsq_final_b_kernel(const float* __restrict__ part, int nb, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] += s;
}
// 4 consecutive parameters per thread: every fp32 state access is one fully coalesced 16-byte lane access (1 KiB per
// wave instruction), the bf16 gradient / copy 8 bytes per lane; two independent chunks per loop trip keep 14 loads in
// flight per thread.
__device__ __forceinline__ void adamw4(float* __restrict__ p, bf16* __restrict__ pb, const bf16* __restrict__ g, float* __restrict__ m,
                                       float* __restrict__ v, int64_t i4, float clip, float lr, float b1, float b2, float eps, float wd,
                                       float bc1, float bc2_sqrt, f32x4 pv, f32x4 mv, f32x4 vv, bf16x4 gv) {
  bf16x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float gi = (float)gv[k] * clip;
    float pi = pv[k] * (1.f - lr * wd);
    const float mi = b1 * mv[k] + (1.f - b1) * gi;
    const float vi = b2 * vv[k] + (1.f - b2) * gi * gi;
    pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    pv[k] = pi; mv[k] = mi; vv[k] = vi; o[k] = (bf16)pi;
  }
  *reinterpret_cast<f32x4*>(p + i4 * 4) = pv;
  *reinterpret_cast<f32x4*>(m + i4 * 4) = mv;
  *reinterpret_cast<f32x4*>(v + i4 * 4) = vv;
  if (pb) *reinterpret_cast<bf16x4*>(pb + i4 * 4) = o;
}
__global__ __launch_bounds__(256) void adamw_g16_kernel(float* __restrict__ p, bf16* __restrict__ pb, const bf16* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n4, float lr, float b1,
                                                        float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                        const float* __restrict__ sumsq, float max_norm) {
  float clip = 1.f;
  if (max_norm > 0.f && sumsq) clip = fminf(1.f, max_norm / (sqrtf(sumsq[0]) + 1e-6f));
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += 2 * stride) {
    const int64_t j = i + stride;
    const bool two = j < n4;
    const f32x4 p0 = *reinterpret_cast<const f32x4*>(p + i * 4), m0 = *reinterpret_cast<const f32x4*>(m + i * 4),
                v0 = *reinterpret_cast<const f32x4*>(v + i * 4);
    const bf16x4 g0 = *reinterpret_cast<const bf16x4*>(g + i * 4);
    f32x4 p1{}, m1{}, v1{}; bf16x4 g1{};
    if (two) {
      p1 = *reinterpret_cast<const f32x4*>(p + j * 4); m1 = *reinterpret_cast<const f32x4*>(m + j * 4);
      v1 = *reinterpret_cast<const f32x4*>(v + j * 4); g1 = *reinterpret_cast<const bf16x4*>(g + j * 4);
    }
    adamw4(p, pb, g, m, v, i, clip, lr, b1, b2, eps, wd, bc1, bc2_sqrt, p0, m0, v0, g0);
    if (two) adamw4(p, pb, g, m, v, j, clip, lr, b1, b2, eps, wd, bc1, bc2_sqrt, p1, m1, v1, g1);
  }
}

// ---------------------------------------------------------------- activations backward
__device__ __forceinline__ float act_grad(float x, int act) {
  if (act == VLB_ACT_SILU) { const float s = sigmoid_f(x); return s * (1.f + x * (1.f - s)); }
  if (act == VLB_ACT_GELU) return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
  if (act == VLB_ACT_QUICK_GELU) { const float s = sigmoid_f(1.702f * x); return s * (1.f + 1.702f * x * (1.f - s)); }
  return 1.f;
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, bf16* __restrict__ dx,
                                                      int64_t n8, int act) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float xv[8], gv[8], o[8]; ld8(x + i * 8, xv); ld8(dy + i * 8, gv);
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = gv[k] * act_grad(xv[k], act);
  st8(dx + i * 8, o);
}

__device__ __forceinline__ float act_val(float x, int act) {
  if (act == VLB_ACT_SILU) return silu_f(x);
  if (act == VLB_ACT_GELU) return gelu_erf_f(x);
  if (act == VLB_ACT_QUICK_GELU) return quick_gelu_f(x);
  return x;
}
__global__ __launch_bounds__(256) void act_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int64_t n8, int act) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float xv[8], o[8]; ld8(x + i * 8, xv);
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = act_val(xv[k], act);
  st8(y + i * 8, o);
}

// ---------------------------------------------------------------- LayerNorm (+ residual, + activation) backward
// forward: z = LN(x; w, b) + res ; y = act(z).  Given dy: dz = dy * act'(z) ; d res = dz ; dx = LN'(dz) ;
// d gamma / d beta partial sums per block of rows (reduced in fixed order by colpart_reduce_kernel).
// One wave per row, 4 waves per block, rows_per_block rows per block.
template <int NI>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w, const bf16* __restrict__ b,
                                                            const bf16* __restrict__ res, const bf16* __restrict__ dy,
                                                            bf16* __restrict__ dx, bf16* __restrict__ dres, float* __restrict__ part_g,
                                                            float* __restrict__ part_b, int rows, int dim, float eps, int act,
                                                            int rows_per_block) {
  extern __shared__ float red[];                      // [2][4][dim]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float ag[NI][8], ab[NI][8];
#pragma unroll
  for (int k = 0; k < NI; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) { ag[k][i] = 0.f; ab[k][i] = 0.f; }
  const int rbeg = blockIdx.x * rows_per_block, rend = min(rows, rbeg + rows_per_block);
  for (int row = rbeg + wave; row < rend; row += 4) {
    const int64_t off = (int64_t)row * dim;
    float xv[NI][8], gv[NI][8], wv[NI][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int c = lane * 8 + k * 512;
      if (c < dim) { ld8(x + off + c, xv[k]); ld8(dy + off + c, gv[k]); ld8(w + c, wv[k]); }
      else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { xv[k][i] = 0.f; gv[k][i] = 0.f; wv[k][i] = 0.f; }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) s += xv[k][i];
    }
    const float mean = wave_sum(s) / dim;
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int c = lane * 8 + k * 512;
      if (c < dim)
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = xv[k][i] - mean; ss += d * d; }
    }
    const float rstd = rsqrtf(wave_sum(ss) / dim + eps);
    // dz (through the activation), and the two row sums LayerNorm's backward needs
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int c = lane * 8 + k * 512;
      if (c >= dim) continue;
      float bv[8], rv[8];
      ld8(b + c, bv);
      if (res) ld8(res + off + c, rv);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh = (xv[k][i] - mean) * rstd;
        float dz = gv[k][i];
        if (act != VLB_ACT_NONE) {
          // pre-activation recomputed exactly as the forward kernel forms it (fp32, no intermediate rounding)
          float z = xh * wv[k][i] + bv[i];
          if (res) z += rv[i];
          dz *= act_grad(z, act);
        }
        gv[k][i] = dz;
        xv[k][i] = xh;
        ag[k][i] += dz * xh;
        ab[k][i] += dz;
        s1 += dz * wv[k][i];
        s2 += dz * wv[k][i] * xh;
      }
      if (dres) st8(dres + off + c, gv[k]);
    }
    s1 = wave_sum(s1) / dim; s2 = wave_sum(s2) / dim;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int c = lane * 8 + k * 512;
      if (c >= dim) continue;
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = rstd * (gv[k][i] * wv[k][i] - s1 - xv[k][i] * s2);
      st8(dx + off + c, o);
    }
  }
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int c = lane * 8 + k * 512;
    if (c < dim)
#pragma unroll
      for (int i = 0; i < 8; ++i) { red[wave * dim + c + i] = ag[k][i]; red[(4 + wave) * dim + c + i] = ab[k][i]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < dim; c += 256) {
    part_g[(int64_t)blockIdx.x * dim + c] = red[c] + red[dim + c] + red[2 * dim + c] + red[3 * dim + c];
    part_b[(int64_t)blockIdx.x * dim + c] = red[4 * dim + c] + red[5 * dim + c] + red[6 * dim + c] + red[7 * dim + c];
  }
}

// ---------------------------------------------------------------- depthwise 3x3 weight gradient
// x, dy: [N, H, W, C] channels-last; dw9[tap][c] = sum_{n,h,w} dy[n,h,w,c] * x[n,h+dh,w+dw,c] (zero padding).
// Block = 256 channels-octets?  thread owns 8 channels; blockIdx.y = image; partial [N][9][C] reduced afterwards.
__global__ __launch_bounds__(256) void dwconv_dw_partial_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, float* __restrict__ part,
                                                                int H, int W, int C) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (c >= C) return;
  const int n = blockIdx.y / H, h = blockIdx.y % H;       // one block row per (image, image row): N*H partial slabs
  float acc[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[t][i] = 0.f;
  const bf16* xn = x + (int64_t)n * H * W * C;
  const bf16* gn = dy + (int64_t)n * H * W * C;
  for (int w = 0; w < W; ++w) {
    float g[8]; ld8(gn + ((int64_t)h * W + w) * C + c, g);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
      if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
      float v[8]; ld8(xn + ((int64_t)hh * W + ww) * C + c, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[t][i] += g[i] * v[i];
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 8; ++i) part[((int64_t)blockIdx.y * 9 + t) * C + c + i] = acc[t][i];
}

// ---------------------------------------------------------------- squeeze-excite backward
// forward: y[n,p,c] = x[n,p,c] * sigmoid(s[n,c]).  dgate[n,c] = (sum_p dy*x) * g(1-g)  (d of the gate LOGIT s)
__global__ __launch_bounds__(256) void se_bwd_gate_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, const bf16* __restrict__ s,
                                                          bf16* __restrict__ ds, int HW, int C) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (c >= C) return;
  const int n = blockIdx.y;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int p = 0; p < HW; ++p) {
    float xv[8], gv[8];
    ld8(x + ((int64_t)n * HW + p) * C + c, xv); ld8(dy + ((int64_t)n * HW + p) * C + c, gv);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += xv[i] * gv[i];
  }
  float sv[8], o[8]; ld8(s + (int64_t)n * C + c, sv);
#pragma unroll
  for (int i = 0; i < 8; ++i) { const float g = sigmoid_f(sv[i]); o[i] = acc[i] * g * (1.f - g); }
  st8(ds + (int64_t)n * C + c, o);
}
// dx[n,p,c] = dy * sigmoid(s[n,c]) + dpool[n,c] / HW      (dpool: gradient of the squeeze mean, may be null)
__global__ __launch_bounds__(256) void se_bwd_x_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ s, const bf16* __restrict__ dpool,
                                                       bf16* __restrict__ dx, int HW, int C, int64_t total8) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total8) return;
  const int c8 = C / 8;
  const int c = (int)(i % c8) * 8;
  const int64_t n = i / c8 / HW;
  float gv[8], sv[8], o[8], pv[8];
  ld8(dy + i * 8, gv); ld8(s + n * C + c, sv);
  if (dpool) ld8(dpool + n * C + c, pv);
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = gv[k] * sigmoid_f(sv[k]) + (dpool ? pv[k] / HW : 0.f);
  st8(dx + i * 8, o);
}

// ---------------------------------------------------------------- inverse of im2col3d (k=2, s=2, p=1)
// cols: [B*T2*H2*W2, 8*C] with tap order (kt,kh,kw); dx[b,t,h,w,c] = dcols[window(t,h,w)][tap(t,h,w)][c]
__global__ __launch_bounds__(256) void col2im3d_kernel(const bf16* __restrict__ dcols, bf16* __restrict__ dx, int T, int H, int W, int C,
                                                       int T2, int H2, int W2, int64_t total8) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total8) return;
  const int c8 = C / 8;
  const int c = (int)(i % c8) * 8;
  int64_t r = i / c8;
  const int w = r % W; r /= W;
  const int h = r % H; r /= H;
  const int t = r % T; const int64_t b = r / T;
  const int pt = t + 1, ph = h + 1, pw = w + 1;           // padded coordinates
  const int64_t win = ((b * T2 + pt / 2) * H2 + ph / 2) * W2 + pw / 2;
  const int tap = ((pt & 1) * 2 + (ph & 1)) * 2 + (pw & 1);
  *reinterpret_cast<bf16x8*>(dx + i * 8) = *reinterpret_cast<const bf16x8*>(dcols + (win * 8 + tap) * C + c);
}

inline int blocks_for(int64_t n, int block) { return (int)((n + block - 1) / block); }
}  // namespace

extern "C" int vlb_transpose_pad(const void* in, int ld_in, void* out, int ld_out, int R, int C, int Rpad, void* stream) {
  VLB_REQUIRE(in && out && R > 0 && C > 0 && Rpad >= R && Rpad % 8 == 0 && ld_out >= Rpad && ld_out % 8 == 0 && ld_in >= C,
              "transpose_pad: bad shape R=%d C=%d Rpad=%d ld_in=%d ld_out=%d", R, C, Rpad, ld_in, ld_out);
  VLB_REQUIRE((((uintptr_t)in | (uintptr_t)out) % 16) == 0 && ld_in % 8 == 0, "transpose_pad: 16-byte alignment required");
  dim3 grid((C + 63) / 64, (Rpad + 63) / 64);
  hipLaunchKernelGGL(transpose_pad_kernel, grid, dim3(256), 0, as_stream(stream), (const bf16*)in, ld_in, (bf16*)out, ld_out, R, C, Rpad);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

// rows of a norm backward handled by one block of the partial pass (=> number of partial slabs)
static inline int norm_rows_per_block(int rows) { int r = (rows + 255) / 256; return r < 4 ? 4 : (r + 3) / 4 * 4; }

extern "C" int64_t vlb_norm_bwd_ws_floats(int rows, int dim) {
  const int rpb = norm_rows_per_block(rows);
  return 2ll * ((rows + rpb - 1) / rpb) * dim;
}

extern "C" int vlb_rmsnorm_bwd_dw(const void* x, const void* dy, void* dw_bf16, float* ws, int rows, int dim, float eps, void* stream) {
  VLB_REQUIRE(x && dy && dw_bf16 && ws && rows > 0 && dim % 8 == 0 && dim <= 4096, "rmsnorm_bwd_dw: bad args (dim <= 4096)");
  hipStream_t st = as_stream(stream);
  const int rpb = norm_rows_per_block(rows), nb = (rows + rpb - 1) / rpb;
  const int lds = 4 * dim * (int)sizeof(float);
#define VLB_RDW(NI) hipLaunchKernelGGL(rmsnorm_dw_partial_kernel<NI>, dim3(nb), dim3(256), lds, st, (const bf16*)x, (const bf16*)dy, ws, rows, dim, eps, rpb)
  if (dim <= 512) VLB_RDW(1); else if (dim <= 1024) VLB_RDW(2); else if (dim <= 2048) VLB_RDW(4); else VLB_RDW(8);
#undef VLB_RDW
  VLB_LAUNCH_CHECK();
  hipLaunchKernelGGL(colpart_reduce_kernel, dim3((dim + 31) / 32), dim3(256), 0, st, ws, nb, dim, (bf16*)dw_bf16, (float*)nullptr);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

// rows per block of the fused backward: enough blocks to fill the chip several times over (a memory-bound sweep)
static inline int rms_full_rows_per_block(int rows) { int r = (rows + 1023) / 1024; return r < 4 ? 4 : (r + 3) / 4 * 4; }
extern "C" int64_t vlb_rmsnorm_bwd_full_ws_floats(int rows, int dim) {
  const int rpb = rms_full_rows_per_block(rows);
  return (int64_t)((rows + rpb - 1) / rpb) * dim;
}
extern "C" int vlb_rmsnorm_bwd_full(const void* x, const void* w, const void* dy, const void* dx_in, void* dx, void* dw_bf16, float* ws,
                                    int rows, int dim, float eps, void* stream) {
  VLB_REQUIRE(x && w && dy && dx && dw_bf16 && ws && rows > 0 && dim % 8 == 0 && dim <= 4096, "rmsnorm_bwd_full: bad args (dim <= 4096)");
  hipStream_t st = as_stream(stream);
  const int rpb = rms_full_rows_per_block(rows), nb = (rows + rpb - 1) / rpb;
  const int lds = 4 * dim * (int)sizeof(float);
#define VLB_RBF(NI) hipLaunchKernelGGL(rmsnorm_dw_partial_kernel<NI>, dim3(nb), dim3(256), lds, st, (const bf16*)x, (const bf16*)dy, ws, rows, dim, eps, rpb, \
                                       (const bf16*)w, (const bf16*)dx_in, (bf16*)dx)
  if (dim <= 512) VLB_RBF(1); else if (dim <= 1024) VLB_RBF(2); else if (dim <= 2048) VLB_RBF(4); else VLB_RBF(8);
#undef VLB_RBF
  VLB_LAUNCH_CHECK();
  hipLaunchKernelGGL(colpart_reduce_kernel, dim3((dim + 31) / 32), dim3(256), 0, st, ws, nb, dim, (bf16*)dw_bf16, (float*)nullptr);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

extern "C" int vlb_layernorm_bwd(const void* x, const void* w, const void* b, const void* residual, const void* dy, void* dx, void* dres,
                                 void* dw_bf16, void* db_bf16, float* ws, int rows, int dim, float eps, int act, void* stream) {
  VLB_REQUIRE(x && w && b && dy && dx && dw_bf16 && db_bf16 && ws && rows > 0 && dim % 8 == 0 && dim <= 4096, "layernorm_bwd: bad args (dim <= 4096)");
  hipStream_t st = as_stream(stream);
  const int rpb = norm_rows_per_block(rows), nb = (rows + rpb - 1) / rpb;
  float* pg = ws; float* pb = ws + (int64_t)nb * dim;
  const int lds = 8 * dim * (int)sizeof(float);
#define VLB_LNB(NI)                                                                                                              \
  do {                                                                                                                           \
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&layernorm_bwd_kernel<NI>),                 \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 8 * NI * 512 * (int)sizeof(float)); \
    if (attr != hipSuccess) { vlb_set_error("layernorm_bwd: LDS reservation failed"); return VLB_ERR_LAUNCH; }                   \
    hipLaunchKernelGGL(layernorm_bwd_kernel<NI>, dim3(nb), dim3(256), lds, st, (const bf16*)x, (const bf16*)w, (const bf16*)b,   \
                       (const bf16*)residual, (const bf16*)dy, (bf16*)dx, (bf16*)dres, pg, pb, rows, dim, eps, act, rpb);        \
  } while (0)
  if (dim <= 512) VLB_LNB(1); else if (dim <= 1024) VLB_LNB(2); else if (dim <= 2048) VLB_LNB(4); else VLB_LNB(8);
#undef VLB_LNB
  VLB_LAUNCH_CHECK();
  hipLaunchKernelGGL(colpart_reduce_kernel, dim3((dim + 31) / 32), dim3(256), 0, st, pg, nb, dim, (bf16*)dw_bf16, (float*)nullptr);
  hipLaunchKernelGGL(colpart_reduce_kernel, dim3((dim + 31) / 32), dim3(256), 0, st, pb, nb, dim, (bf16*)db_bf16, (float*)nullptr);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

extern "C" int vlb_act_bwd(const void* x, const void* dy, void* dx, int64_t n, int act, void* stream) {
  VLB_REQUIRE(x && dy && dx && n > 0 && n % 8 == 0, "act_bwd: n must be a positive multiple of 8");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks_for(n / 8, 256)), dim3(256), 0, as_stream(stream), (const bf16*)x, (const bf16*)dy,
                     (bf16*)dx, n / 8, act);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

extern "C" int vlb_act_fwd(const void* x, void* y, int64_t n, int act, void* stream) {
  VLB_REQUIRE(x && y && n > 0 && n % 8 == 0, "act_fwd: n must be a positive multiple of 8");
  hipLaunchKernelGGL(act_fwd_kernel, dim3(blocks_for(n / 8, 256)), dim3(256), 0, as_stream(stream), (const bf16*)x, (bf16*)y, n / 8, act);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

extern "C" int64_t vlb_colsum_ws_floats(int rows, int dim) {
  const int rpb = norm_rows_per_block(rows);
  return (int64_t)((rows + rpb - 1) / rpb) * dim;
}
extern "C" int vlb_colsum(const void* x, int ld, void* out_bf16, float* ws, int rows, int dim, void* stream) {
  VLB_REQUIRE(x && out_bf16 && ws && rows > 0 && dim % 8 == 0 && ld % 8 == 0 && ld >= dim, "colsum: bad args");
  hipStream_t st = as_stream(stream);
  const int rpb = norm_rows_per_block(rows), nb = (rows + rpb - 1) / rpb;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((dim / 8 + 255) / 256, nb), dim3(256), 0, st, (const bf16*)x, ld, ws, rows, dim, rpb);
  VLB_LAUNCH_CHECK();
  hipLaunchKernelGGL(colpart_reduce_kernel, dim3((dim + 31) / 32), dim3(256), 0, st, ws, nb, dim, (bf16*)out_bf16, (float*)nullptr);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

extern "C" int vlb_embed_grad(const void* d_embeds, int ld, const int* tok, const int* beg, const int* rows, int n_tokens, void* dE, int D,
                              void* stream) {
  VLB_REQUIRE(d_embeds && tok && beg && rows && dE && n_tokens > 0 && D % 8 == 0 && ld % 8 == 0, "embed_grad: bad args");
  hipLaunchKernelGGL(embed_grad_kernel, dim3(n_tokens), dim3(256), 0, as_stream(stream), (const bf16*)d_embeds, ld, tok, beg, rows,
                     (bf16*)dE, D);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

extern "C" int vlb_grad_sumsq_bf16(const void* g, int64_t n, float* sumsq, float* ws, void* stream) {
  VLB_REQUIRE(n > 0 && n % 8 == 0 && g && sumsq && ws && ((uintptr_t)g % 16) == 0, "grad_sumsq_bf16: n must be a positive multiple of 8");
  hipStream_t st = as_stream(stream);
  const int64_t n8 = n / 8;
  int64_t nb64 = (n8 + 255) / 256;
  const int nb = (int)(nb64 > kSumsqBlocksB ? kSumsqBlocksB : nb64);
  hipLaunchKernelGGL(sumsq_bf16_kernel, dim3(nb), dim3(256), 0, st, (const bf16*)g, n8, ws);
  VLB_LAUNCH_CHECK();
  hipLaunchKernelGGL(sumsq_final_b_kernel, dim3(1), dim3(256), 0, st, ws, nb, sumsq);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

extern "C" int vlb_adamw_step_g16(float* master, void* param_bf16, const void* grad_bf16, float* m, float* v, int64_t n, float lr,
                                  float beta1, float beta2, float eps, float weight_decay, int step, const float* sumsq, float max_norm,
                                  void* stream) {
  VLB_REQUIRE(n > 0 && n % 8 == 0 && master && grad_bf16 && m && v && step >= 1, "adamw_g16: n must be a positive multiple of 8");
  VLB_REQUIRE((((uintptr_t)master | (uintptr_t)m | (uintptr_t)v) % 16) == 0 && (((uintptr_t)grad_bf16 | (uintptr_t)param_bf16) % 8) == 0,
              "adamw_g16: buffers must be 16-byte (fp32) / 8-byte (bf16) aligned");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  const int64_t n4 = n / 4;
  int64_t nb = (n4 + 511) / 512; if (nb > 32768) nb = 32768;
  hipLaunchKernelGGL(adamw_g16_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), master, (bf16*)param_bf16, (const bf16*)grad_bf16,
                     m, v, n4, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, sumsq, max_norm);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

extern "C" int64_t vlb_dwconv3x3_bwd_w_ws_floats(int N, int H, int C) { return (int64_t)N * H * 9 * C; }
extern "C" int vlb_dwconv3x3_bwd_w(const void* x, const void* dy, void* dw9_bf16, float* ws, int N, int H, int W, int C, void* stream) {
  VLB_REQUIRE(x && dy && dw9_bf16 && ws && N > 0 && H > 0 && W > 0 && C % 8 == 0, "dwconv3x3_bwd_w: bad args");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(dwconv_dw_partial_kernel, dim3((C / 8 + 255) / 256, N * H), dim3(256), 0, st, (const bf16*)x, (const bf16*)dy, ws, H, W, C);
  VLB_LAUNCH_CHECK();
  hipLaunchKernelGGL(colpart_reduce_kernel, dim3((9 * C + 31) / 32), dim3(256), 0, st, ws, N * H, 9 * C, (bf16*)dw9_bf16, (float*)nullptr);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

extern "C" int vlb_se_bwd_gate(const void* x, const void* dy, const void* s, void* ds, int N, int HW, int C, void* stream) {
  VLB_REQUIRE(x && dy && s && ds && N > 0 && HW > 0 && C % 8 == 0, "se_bwd_gate: bad args");
  hipLaunchKernelGGL(se_bwd_gate_kernel, dim3((C / 8 + 255) / 256, N), dim3(256), 0, as_stream(stream), (const bf16*)x, (const bf16*)dy,
                     (const bf16*)s, (bf16*)ds, HW, C);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
extern "C" int vlb_se_bwd_x(const void* dy, const void* s, const void* dpool, void* dx, int N, int HW, int C, void* stream) {
  VLB_REQUIRE(dy && s && dx && N > 0 && HW > 0 && C % 8 == 0, "se_bwd_x: bad args");
  const int64_t total8 = (int64_t)N * HW * (C / 8);
  hipLaunchKernelGGL(se_bwd_x_kernel, dim3(blocks_for(total8, 256)), dim3(256), 0, as_stream(stream), (const bf16*)dy, (const bf16*)s,
                     (const bf16*)dpool, (bf16*)dx, HW, C, total8);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

extern "C" int vlb_col2im3d_k2s2p1(const void* dcols, void* dx, int B, int T, int H, int W, int C, void* stream) {
  VLB_REQUIRE(dcols && dx && B > 0 && T > 0 && H > 0 && W > 0 && C % 8 == 0, "col2im3d: bad args");
  const int T2 = T / 2 + 1, H2 = H / 2 + 1, W2 = W / 2 + 1;
  const int64_t total8 = (int64_t)B * T * H * W * (C / 8);
  hipLaunchKernelGGL(col2im3d_kernel, dim3(blocks_for(total8, 256)), dim3(256), 0, as_stream(stream), (const bf16*)dcols, (bf16*)dx, T, H, W,
                     C, T2, H2, W2, total8);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
