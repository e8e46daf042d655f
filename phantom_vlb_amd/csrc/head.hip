// Fused brain head (the arithmetic the reference itself owns):
//   hidden --LN1--> sum_s w[b,s]*(.) --LN2--> dropout --Linear(E->V)--> pred ; loss = MSE + lambda*||W||^2
// reference: src/litmodule/videollama2_vlb_litmodule.py:245-254,302 ; src/utils.py:56,66-71.
//
// HBM-bound.  The LN1-normalised [B,S,E] tensor is never materialised: because LN1's affine is
// per-column,   pooled[e] = g1[e] * sum_s w_s*rstd_s*(x_se - mu_s) + b1[e] * sum_s w_s,
// so one pass over `hidden` with per-token (mu, rstd) suffices, and tokens whose HRF weight is zero
// (the prompt, instruction and padding spans) are skipped altogether.
// All cross-block reductions go through partial slabs summed in a fixed order (bitwise reproducible).
#include "common.hpp"

namespace {

constexpr int POOL_ROWS = 32;     // tokens per pooling block (8 waves x 4 tokens)
constexpr int RIDGE_ROWS = 16;    // ridge rows per block (4 waves x 4 rows)
constexpr int BMAX = 8;           // clips handled per pass by the ridge kernels
constexpr int DZ_SPLIT = 32;      // V-splits of the dz reduction

__device__ __forceinline__ void ld8(const bf16* p, float (&v)[8]) {
  const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
}

// ---------------------------------------------------------------- forward 1: LN1 stats + weighted pool
template <int NI>   // NI = ceil(E/512) chunks of 8 columns per lane
__global__ __launch_bounds__(512) void head_pool_kernel(const bf16* __restrict__ hidden, const float* __restrict__ wmask,
                                                        float* __restrict__ partial, float* __restrict__ stats, int S,
                                                        int E, float eps, const int* __restrict__ cu) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [4][E]
  const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
  // packed hidden: clip b = rows [cu[b], cu[b+1]); wmask / stats stay dense [B,S]
  const int Sb = cu ? cu[b + 1] - cu[b] : S;
  const int64_t row0 = cu ? cu[b] : (int64_t)b * S;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[NI][8];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
  float sw = 0.f;
  // this wave's tokens: blk*32 + wave + {0,8,16,24}; rows with zero weight are skipped (wave-uniform).
  // Software pipeline: the next live row is in flight (raw bf16) while the current one is reduced.
  auto load_row = [&](int s_, bf16x8 (&dst)[NI]) {
    const bf16* xr = hidden + (row0 + s_) * E;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane * 8 + i * 512;
      dst[i] = c < E ? *reinterpret_cast<const bf16x8*>(xr + c) : bf16x8{};
    }
  };
  int live[POOL_ROWS / 8];
  float wl[POOL_ROWS / 8];
  int nlive = 0;
#pragma unroll
  for (int q = 0; q < POOL_ROWS / 8; ++q) {
    const int s_ = blk * POOL_ROWS + wave + 8 * q;
    const float w_ = s_ < Sb ? wmask[(int64_t)b * S + s_] : 0.f;
    if (w_ != 0.f) { live[nlive] = s_; wl[nlive] = w_; ++nlive; }
  }
  bf16x8 cur[NI], nxt[NI];
  if (nlive > 0) load_row(live[0], cur);
#pragma unroll
  for (int q = 0; q < POOL_ROWS / 8; ++q) {
    if (q >= nlive) break;
    if (q + 1 < nlive) load_row(live[q + 1], nxt);
    const int s = live[q];
    const float w = wl[q];
    float x[NI][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) { x[i][j] = (float)cur[i][j]; sum += x[i][j]; }
    const float mu = wave_sum(sum) / E;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane * 8 + i * 512;
      if (c < E) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = x[i][j] - mu; var += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(var) / E + eps);
    const float a = w * rstd;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] += a * (x[i][j] - mu);
    sw += w;
    if (lane == 0) { stats[((int64_t)b * S + s) * 2] = mu; stats[((int64_t)b * S + s) * 2 + 1] = rstd; }
#pragma unroll
    for (int i = 0; i < NI; ++i) cur[i] = nxt[i];
  }
  // deterministic tree reduction over the 8 waves through LDS
  for (int stride = 4; stride >= 1; stride >>= 1) {
    if (wave >= stride && wave < 2 * stride) {
      float* dst = red + (wave - stride) * E;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = lane * 8 + i * 512;
        if (c < E) {
          *reinterpret_cast<f32x4*>(dst + c) = f32x4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
          *reinterpret_cast<f32x4*>(dst + c + 4) = f32x4{acc[i][4], acc[i][5], acc[i][6], acc[i][7]};
        }
      }
      if (lane == 0) red[4 * E + (wave - stride)] = sw;
    }
    __syncthreads();
    if (wave < stride) {
      const float* src = red + wave * E;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = lane * 8 + i * 512;
        if (c < E) {
          const f32x4 lo = *reinterpret_cast<const f32x4*>(src + c), hi = *reinterpret_cast<const f32x4*>(src + c + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { acc[i][j] += lo[j]; acc[i][4 + j] += hi[j]; }
        }
      }
      sw += red[4 * E + wave];
    }
    __syncthreads();
  }
  if (wave == 0) {
    float* dst = partial + ((int64_t)b * nblk + blk) * (E + 2);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane * 8 + i * 512;
      if (c < E) {
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[c + j] = acc[i][j];
      }
    }
    if (lane == 0) dst[E] = sw;
  }
}

// ---------------------------------------------------------------- forward 2a: sum the pooling slabs (fixed order)
// grid (ceil(E/256), B): thread e sums partial[b][0..nblk)[e]; thread 0 of block x=0 sums the weights.
__global__ __launch_bounds__(256) void head_reduce_kernel(const float* __restrict__ partial, int nblk,
                                                          float* __restrict__ pooled_raw, float* __restrict__ sumw, int E) {
  const int b = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
  const float* pb = partial + (int64_t)b * nblk * (E + 2);
  if (e < E) {
    float raw = 0.f;
#pragma unroll 8
    for (int k = 0; k < nblk; ++k) raw += pb[(int64_t)k * (E + 2) + e];
    pooled_raw[(int64_t)b * E + e] = raw;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float sw = 0.f;
    for (int k = 0; k < nblk; ++k) sw += pb[(int64_t)k * (E + 2) + E];
    sumw[b] = sw;
  }
}
// ---------------------------------------------------------------- forward 2b: LN1 affine, LN2, dropout -> z (bf16)
__global__ __launch_bounds__(1024) void head_ln2_kernel(const float* __restrict__ pooled_raw, const float* __restrict__ sumw,
                                                        const bf16* __restrict__ g1, const bf16* __restrict__ b1,
                                                        const bf16* __restrict__ g2, const bf16* __restrict__ b2,
                                                        const float* __restrict__ keep, float* __restrict__ zhat,
                                                        float* __restrict__ ln2_rstd, bf16* __restrict__ z, int E, float eps) {
  __shared__ float red[16];
  const int b = blockIdx.x;
  const float sw = sumw[b];
  float lsum = 0.f;
  for (int e = threadIdx.x; e < E; e += blockDim.x)
    lsum += (float)g1[e] * pooled_raw[(int64_t)b * E + e] + (float)b1[e] * sw;
  const float mean = block_sum(lsum, red) / E;
  float lvar = 0.f;
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    const float pv = (float)g1[e] * pooled_raw[(int64_t)b * E + e] + (float)b1[e] * sw;
    lvar += (pv - mean) * (pv - mean);
  }
  const float rstd = rsqrtf(block_sum(lvar, red) / E + eps);
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    const float pv = (float)g1[e] * pooled_raw[(int64_t)b * E + e] + (float)b1[e] * sw;
    const float zh = (pv - mean) * rstd;
    zhat[(int64_t)b * E + e] = zh;
    float zz = zh * (float)g2[e] + (float)b2[e];
    if (keep) zz *= keep[(int64_t)b * E + e];
    z[(int64_t)b * E + e] = (bf16)zz;
  }
  if (threadIdx.x == 0) ln2_rstd[b] = rstd;
}

// ---------------------------------------------------------------- forward 3: ridge GEMV + loss partials
// block = 4 waves x RIDGE_ROWS/4 rows of W; z (bf16 [<=BMAX, E]) is staged once per block into LDS and
// re-read from there for every row (the W stream, 16-byte lanes, is the only HBM traffic).
__global__ __launch_bounds__(256) void ridge_fwd_kernel(const bf16* __restrict__ W, const bf16* __restrict__ bias,
                                                        const bf16* __restrict__ z, const float* __restrict__ y,
                                                        float* __restrict__ pred, float* __restrict__ lpart, int B, int E,
                                                        int V) {
  extern __shared__ __attribute__((aligned(16))) char zsm[];     // [nb][E] bf16
  __shared__ float red[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mse = 0.f, w2 = 0.f;
  for (int b0 = 0; b0 < B; b0 += BMAX) {
    const int nb = min(BMAX, B - b0);
    __syncthreads();
    for (int i = threadIdx.x * 8; i < nb * E; i += 256 * 8)
      *reinterpret_cast<bf16x8*>(zsm + (int64_t)i * 2) = *reinterpret_cast<const bf16x8*>(z + (int64_t)b0 * E + i);
    __syncthreads();
    for (int rr = 0; rr < RIDGE_ROWS / 4; ++rr) {
      const int v = blockIdx.x * RIDGE_ROWS + wave * (RIDGE_ROWS / 4) + rr;
      if (v >= V) break;
      const bf16* wr = W + (int64_t)v * E;
      float acc[BMAX];
#pragma unroll
      for (int i = 0; i < BMAX; ++i) acc[i] = 0.f;
#pragma unroll 4
      for (int c = lane * 8; c < E; c += 512) {
        float w[8]; ld8(wr + c, w);
        if (b0 == 0) {
#pragma unroll
          for (int j = 0; j < 8; ++j) w2 += w[j] * w[j];
        }
#pragma unroll
        for (int i = 0; i < BMAX; ++i) {
          if (i < nb) {
            float zz[8]; ld8(reinterpret_cast<const bf16*>(zsm) + (int64_t)i * E + c, zz);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i] += w[j] * zz[j];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < BMAX; ++i) {
        if (i < nb) {
          const float pv = wave_sum(acc[i]) + (float)bias[v];
          if (lane == 0) {
            pred[(int64_t)(b0 + i) * V + v] = pv;
            const float d = y ? pv - y[(int64_t)(b0 + i) * V + v] : 0.f;
            mse += d * d;
          }
        }
      }
    }
  }
  w2 = wave_sum(w2);
  if (lane == 0) { red[wave] = mse; red[4 + wave] = w2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    lpart[blockIdx.x * 2] = red[0] + red[1] + red[2] + red[3];
    lpart[blockIdx.x * 2 + 1] = red[4] + red[5] + red[6] + red[7];
  }
}
// ---------------------------------------------------------------- forward 3 (MFMA form, E <= 4096, E % 128 == 0, B <= 16)
// pred[b, v] = sum_e z[b,e] W[v,e]: a skinny contraction (M = B <= 16) -> v_mfma_f32_16x16x32_bf16 with the
// MFMA rows = 16 rows of W and the MFMA columns = clips.  Each of the 4 waves of a block owns one QUARTER
// of E and keeps its z fragments in registers for the whole kernel (<= 128 VGPRs), so the only memory
// stream is W: 32 independent 1-KiB fragment loads per wave per row tile.  Blocks are persistent over
// row tiles; the four K-quarters are summed through LDS; sum(W^2) and the squared error ride along.
template <int KSTEPS>   // E / 128: k-steps of 32 per wave
__global__ __launch_bounds__(256) void ridge_fwd_mfma_kernel(const bf16* __restrict__ W, const bf16* __restrict__ bias,
                                                             const bf16* __restrict__ z, const float* __restrict__ y,
                                                             float* __restrict__ pred, float* __restrict__ lpart, int B,
                                                             int E, int V) {
  __shared__ float red[3][4][64];
  __shared__ float lred[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int kq = wave * KSTEPS * 32;                       // first column of this wave's quarter
  // B operand: lane holds z[b = fr][kq + 32*s + 8*fq .. +8]  (zero rows for b >= B)
  bf16x8 zf[KSTEPS];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    zf[s] = bf16x8{};
    if (fr < B) zf[s] = *reinterpret_cast<const bf16x8*>(z + (int64_t)fr * E + kq + 32 * s + 8 * fq);
  }
  float mse = 0.f, w2 = 0.f;
  const int ntiles = (V + 15) / 16;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int v = min(t * 16 + fr, V - 1);
    const bf16* wr = W + (int64_t)v * E + kq + 8 * fq;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool count = t * 16 + fr < V;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wr + 32 * s);
      if (count) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = (float)wf[j]; w2 += f * f; }
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, zf[s], acc, 0, 0, 0);
    }
    __syncthreads();                                       // red[] free again
    if (wave > 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave - 1][e][lane] = acc[e];
    }
    __syncthreads();
    if (wave == 0 && fr < B) {
      // lane holds clip b = fr, rows v = t*16 + 4*fq + e
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int vv = t * 16 + 4 * fq + e;
        if (vv < V) {
          const float pv = acc[e] + red[0][e][lane] + red[1][e][lane] + red[2][e][lane] + (float)bias[vv];
          pred[(int64_t)fr * V + vv] = pv;
          const float d = y ? pv - y[(int64_t)fr * V + vv] : 0.f;
          mse += d * d;
        }
      }
    }
  }
  mse = wave_sum(mse);
  w2 = wave_sum(w2);
  if (lane == 0) { lred[wave] = mse; lred[4 + wave] = w2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    lpart[blockIdx.x * 2] = lred[0] + lred[1] + lred[2] + lred[3];
    lpart[blockIdx.x * 2 + 1] = lred[4] + lred[5] + lred[6] + lred[7];
  }
}

// one block: thread i sums entries i, i+256, ... in order; the 256 partials are then tree-summed in LDS
// (fixed order -> reproducible)
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ lpart, int nblk, float inv_bv, float lambda,
                                                            float* __restrict__ out) {
  __shared__ double sm[2][256];
  double mse = 0.0, w2 = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) { mse += lpart[2 * i]; w2 += lpart[2 * i + 1]; }
  sm[0][threadIdx.x] = mse; sm[1][threadIdx.x] = w2;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if ((int)threadIdx.x < s2) { sm[0][threadIdx.x] += sm[0][threadIdx.x + s2]; sm[1][threadIdx.x] += sm[1][threadIdx.x + s2]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = (float)(sm[0][0] * inv_bv);
    out[1] = (float)(lambda * sm[1][0]);
    out[2] = out[0] + out[1];
  }
}

// d pred^T as the skinny-wgrad G operand: dpT[v, b] = gscale * (pred[b,v] - y[b,v]) for b < B, 0 for b < 16
__global__ void dpred_t_kernel(const float* __restrict__ pred, const float* __restrict__ y, bf16* __restrict__ dpT, int B,
                               int V, float gscale) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= (int64_t)V * 16) return;
  const int v = i >> 4, b = i & 15;
  dpT[i] = (bf16)(b < B ? gscale * (pred[(int64_t)b * V + v] - y[(int64_t)b * V + v]) : 0.f);
}

// ---------------------------------------------------------------- backward 1: dW, dbias   (z staged in LDS)
__global__ __launch_bounds__(256) void ridge_bwd_w_kernel(const bf16* __restrict__ W, const bf16* __restrict__ z,
                                                          const float* __restrict__ pred, const float* __restrict__ y,
                                                          float* __restrict__ dW, float* __restrict__ dbias, int B, int E,
                                                          int V, float gscale, float l2coef) {
  extern __shared__ __attribute__((aligned(16))) char zsm[];     // [nb][E] bf16
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b0 = 0; b0 < B; b0 += BMAX) {
    const int nb = min(BMAX, B - b0);
    __syncthreads();
    for (int i = threadIdx.x * 8; i < nb * E; i += 256 * 8)
      *reinterpret_cast<bf16x8*>(zsm + (int64_t)i * 2) = *reinterpret_cast<const bf16x8*>(z + (int64_t)b0 * E + i);
    __syncthreads();
    for (int rr = 0; rr < RIDGE_ROWS / 4; ++rr) {
      const int v = blockIdx.x * RIDGE_ROWS + wave * (RIDGE_ROWS / 4) + rr;
      if (v >= V) break;
      float dp[BMAX];
      float db = 0.f;
#pragma unroll
      for (int i = 0; i < BMAX; ++i) {
        dp[i] = i < nb ? gscale * (pred[(int64_t)(b0 + i) * V + v] - y[(int64_t)(b0 + i) * V + v]) : 0.f;
        db += dp[i];
      }
      if (lane == 0) dbias[v] = (b0 == 0 ? 0.f : dbias[v]) + db;
      const bf16* wr = W + (int64_t)v * E;
#pragma unroll 4
      for (int c = lane * 8; c < E; c += 512) {
        float g[8];
        float* o = dW + (int64_t)v * E + c;
        if (b0 == 0) {
          float w[8]; ld8(wr + c, w);
#pragma unroll
          for (int j = 0; j < 8; ++j) g[j] = l2coef * w[j];
        } else {
          const f32x4 lo = *reinterpret_cast<const f32x4*>(o), hi = *reinterpret_cast<const f32x4*>(o + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { g[j] = lo[j]; g[4 + j] = hi[j]; }
        }
#pragma unroll
        for (int i = 0; i < BMAX; ++i) {
          if (i < nb) {
            float zz[8]; ld8(reinterpret_cast<const bf16*>(zsm) + (int64_t)i * E + c, zz);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] += dp[i] * zz[j];
          }
        }
        *reinterpret_cast<f32x4*>(o) = f32x4{g[0], g[1], g[2], g[3]};
        *reinterpret_cast<f32x4*>(o + 4) = f32x4{g[4], g[5], g[6], g[7]};
      }
    }
  }
}
// ---------------------------------------------------------------- backward 2: dz partials over V splits
// grid (ceil(E/512), DZ_SPLIT), block 256 (4 waves share a split's rows); out part[split][B][E]
__global__ __launch_bounds__(256) void ridge_bwd_z_kernel(const bf16* __restrict__ W, const float* __restrict__ pred,
                                                          const float* __restrict__ y, float* __restrict__ part, int B,
                                                          int E, int V, float gscale) {
  __shared__ float red[3][BMAX][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 512 + lane * 8;
  const int per = (V + DZ_SPLIT - 1) / DZ_SPLIT;
  const int v0 = blockIdx.y * per, v1 = min(V, v0 + per);
  for (int b0 = 0; b0 < B; b0 += BMAX) {
    const int nb = min(BMAX, B - b0);
    float acc[BMAX][8];
#pragma unroll
    for (int i = 0; i < BMAX; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    if (c < E) {
      for (int v = v0 + wave; v < v1; v += 4) {
        float w[8]; ld8(W + (int64_t)v * E + c, w);
#pragma unroll
        for (int i = 0; i < BMAX; ++i) {
          if (i < nb) {
            const float dp = gscale * (pred[(int64_t)(b0 + i) * V + v] - y[(int64_t)(b0 + i) * V + v]);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] += dp * w[j];
          }
        }
      }
    }
    __syncthreads();
    if (wave > 0) {
#pragma unroll
      for (int i = 0; i < BMAX; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wave - 1][i][lane * 8 + j] = acc[i][j];
    }
    __syncthreads();
    if (wave == 0 && c < E) {
#pragma unroll
      for (int i = 0; i < BMAX; ++i) {
        if (i < nb) {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            part[((int64_t)blockIdx.y * B + b0 + i) * E + c + j] =
                acc[i][j] + red[0][i][lane * 8 + j] + red[1][i][lane * 8 + j] + red[2][i][lane * 8 + j];
        }
      }
    }
  }
}
// ---------------------------------------------------------------- backward 3a: sum dz partials, apply dropout
__global__ __launch_bounds__(256) void head_dz_reduce_kernel(const float* __restrict__ part, const float* __restrict__ keep,
                                                             float* __restrict__ dz, int B, int E) {
  const int b = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  float d = 0.f;
#pragma unroll 8
  for (int k = 0; k < DZ_SPLIT; ++k) d += part[((int64_t)k * B + b) * E + e];
  if (keep) d *= keep[(int64_t)b * E + e];
  dz[(int64_t)b * E + e] = d;
}
__global__ __launch_bounds__(256) void head_dz_from16_kernel(const float* __restrict__ dz16, const float* __restrict__ keep,
                                                             float* __restrict__ dz, int E) {
  const int b = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  float d = dz16[(int64_t)b * E + e];
  if (keep) d *= keep[(int64_t)b * E + e];
  dz[(int64_t)b * E + e] = d;
}
// ---------------------------------------------------------------- backward 3b: LN2 backward per clip -> dpooled (in place of draw ws)
__global__ __launch_bounds__(1024) void head_ln2_bwd_kernel(const float* __restrict__ dz, const bf16* __restrict__ g2,
                                                            const float* __restrict__ zhat, const float* __restrict__ ln2_rstd,
                                                            float* __restrict__ dpooled, int E) {
  __shared__ float red[16];
  const int b = blockIdx.x;
  float s1 = 0.f, s2 = 0.f;
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    const float g = dz[(int64_t)b * E + e] * (float)g2[e];
    s1 += g; s2 += g * zhat[(int64_t)b * E + e];
  }
  const float m1 = block_sum(s1, red) / E;
  const float m2 = block_sum(s2, red) / E;
  const float rstd = ln2_rstd[b];
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    const float g = dz[(int64_t)b * E + e] * (float)g2[e];
    dpooled[(int64_t)b * E + e] = rstd * (g - m1 - zhat[(int64_t)b * E + e] * m2);     // d loss / d pooled[b,e]
  }
}
// ---------------------------------------------------------------- backward 3c: norm-parameter grads (sum over clips), draw
__global__ __launch_bounds__(256) void head_param_grads_kernel(const float* __restrict__ dz, float* __restrict__ dpooled,
                                                               const bf16* __restrict__ g1, const float* __restrict__ pooled_raw,
                                                               const float* __restrict__ sumw, const float* __restrict__ zhat,
                                                               float* __restrict__ dg2, float* __restrict__ db2,
                                                               float* __restrict__ dg1, float* __restrict__ db1, int B, int E) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  float a2 = 0.f, c2 = 0.f, a1 = 0.f, c1 = 0.f;
  const float gg = (float)g1[e];
  for (int b = 0; b < B; ++b) {
    const float d = dz[(int64_t)b * E + e], dp = dpooled[(int64_t)b * E + e];
    a2 += d * zhat[(int64_t)b * E + e];
    c2 += d;
    a1 += dp * pooled_raw[(int64_t)b * E + e];
    c1 += dp * sumw[b];
    dpooled[(int64_t)b * E + e] = dp * gg;           // becomes d loss / d pooled_raw[b,e]
  }
  dg2[e] = a2; db2[e] = c2; dg1[e] = a1; db1[e] = c1;
}
// ---------------------------------------------------------------- backward 4: d hidden (one wave per token)
__global__ __launch_bounds__(256) void head_dhidden_kernel(const bf16* __restrict__ hidden, const float* __restrict__ wmask,
                                                           const float* __restrict__ stats, const float* __restrict__ draw,
                                                           bf16* __restrict__ dh, int S, int E, int64_t rows, int B,
                                                           const int* __restrict__ cu) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  int64_t b = row / S, dense = row;             // dense = index into the [B,S] mask / stats
  if (cu) {
    int bb = 0;
    while (bb + 1 < B && row >= cu[bb + 1]) ++bb;
    b = bb; dense = b * S + (row - cu[bb]);
  }
  const float w = wmask[dense];
  bf16* dr = dh + row * E;
  if (w == 0.f) {
    for (int c = lane * 8; c < E; c += 512) *reinterpret_cast<bf16x8*>(dr + c) = bf16x8{};
    return;
  }
  const float mu = stats[dense * 2], rstd = stats[dense * 2 + 1];
  const bf16* xr = hidden + row * E;
  const float* u = draw + b * E;
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane * 8; c < E; c += 512) {
    float x[8]; ld8(xr + c, x);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float g = w * u[c + j]; s1 += g; s2 += g * (x[j] - mu) * rstd; }
  }
  const float m1 = wave_sum(s1) / E, m2 = wave_sum(s2) / E;
  for (int c = lane * 8; c < E; c += 512) {
    float x[8]; ld8(xr + c, x);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)(rstd * (w * u[c + j] - m1 - (x[j] - mu) * rstd * m2));
    *reinterpret_cast<bf16x8*>(dr + c) = o;
  }
}

// ---------------------------------------------------------------- standalone HRFConvolveLayer (src/utils.py:44-56)
// out[b,e] = sum_s w[b,s] * x[b,s,e]: one pass over x, HBM-bound.  grid (HRF_SPLITS, ceil(E/512), B), 4 waves per block; a wave
// walks every 4th token of its split (tokens with zero weight are skipped), a lane owns 8 columns; the four waves are summed
// through LDS and the splits through fp32 slabs in a fixed order (reproducible).  T = bf16 or float embeddings.
constexpr int HRF_SPLITS = 32;

template <typename T>
__global__ __launch_bounds__(256) void hrf_pool_kernel(const T* __restrict__ x, const float* __restrict__ w, float* __restrict__ slab,
                                                       int S, int E) {
  __shared__ float red[3][512];
  const int b = blockIdx.z, c = blockIdx.y * 512 + (threadIdx.x & 63) * 8, wave = threadIdx.x >> 6;
  const int per = (S + HRF_SPLITS - 1) / HRF_SPLITS, s0 = blockIdx.x * per, s1 = min(S, s0 + per);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < E) {
    for (int s = s0 + wave; s < s1; s += 4) {
      const float ws_ = w[(int64_t)b * S + s];
      if (ws_ == 0.f) continue;
      const T* xr = x + ((int64_t)b * S + s) * E + c;
      float v[8];
      if constexpr (sizeof(T) == 2) {
        ld8(reinterpret_cast<const bf16*>(xr), v);
      } else {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(xr), hi = *reinterpret_cast<const f32x4*>(xr + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += ws_ * v[j];
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) red[wave - 1][(threadIdx.x & 63) * 8 + j] = acc[j];
  }
  __syncthreads();
  if (wave == 0 && c < E) {
    float* dst = slab + ((int64_t)b * HRF_SPLITS + blockIdx.x) * E + c;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = (threadIdx.x & 63) * 8 + j;
      dst[j] = ((acc[j] + red[0][k]) + red[1][k]) + red[2][k];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void hrf_pool_reduce_kernel(const float* __restrict__ slab, T* __restrict__ out, int E) {
  const int b = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  float a = 0.f;
  for (int k = 0; k < HRF_SPLITS; ++k) a += slab[((int64_t)b * HRF_SPLITS + k) * E + e];
  out[(int64_t)b * E + e] = (T)a;
}

int reserve_ridge_lds(int bytes) {
  static int reserved = 0;
  if (bytes > reserved) {
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&ridge_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&ridge_bwd_w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e1 != hipSuccess || e2 != hipSuccess) { vlb_set_error("head: cannot reserve %d bytes of LDS for z", bytes); return VLB_ERR_LAUNCH; }
    reserved = bytes;
  }
  return VLB_OK;
}

template <int NI>
int launch_pool(const bf16* hidden, const float* wmask, float* partial, float* stats, int B, int S, int E, float eps,
                const int* cu, hipStream_t st) {
  const int lds = (4 * E + 8) * sizeof(float);
  // once per process and kernel; a function-local static's initialisation is thread-safe (C++11)
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&head_pool_kernel<NI>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (4 * NI * 512 + 8) * (int)sizeof(float));
  if (attr != hipSuccess) { vlb_set_error("head: LDS reservation failed: %s", hipGetErrorString(attr)); return VLB_ERR_LAUNCH; }
  dim3 grid((S + POOL_ROWS - 1) / POOL_ROWS, B);
  hipLaunchKernelGGL((head_pool_kernel<NI>), grid, dim3(512), lds, st, hidden, wmask, partial, stats, S, E, eps, cu);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
}  // namespace

extern "C" int vlb_head_partial_rows(int S) { return (S + POOL_ROWS - 1) / POOL_ROWS; }
extern "C" int64_t vlb_head_ws_floats(int B, int S, int E, int V) {
  const int64_t pool = (int64_t)B * vlb_head_partial_rows(S) * (E + 2);
  const int64_t ridge = 2 * (int64_t)((V + RIDGE_ROWS - 1) / RIDGE_ROWS);
  const int64_t dz = (int64_t)DZ_SPLIT * B * E;
  const int64_t dz_mfma = (int64_t)vlb_wgrad_splits(V) * 16 * E + (int64_t)16 * E + ((int64_t)V * 16 + 1) / 2 + 64;
  int64_t m = pool + ridge;
  if (dz > m) m = dz;
  return dz_mfma > m ? dz_mfma : m;
}

extern "C" int vlb_head_fwd(const void* hidden, const float* wmask, const void* ln1_w, const void* ln1_b,
                            const void* ln2_w, const void* ln2_b, const void* ridge_w, const void* ridge_b,
                            const float* y, const float* keep_scale, float* ws, float* stats, float* pooled_raw,
                            float* sumw, float* zhat, float* ln2_rstd, void* z, float* pred, float* loss_terms, int B,
                            int S, int E, int V, float eps, float l2_lambda, const int* cu_rows, void* stream) {
  VLB_REQUIRE(hidden && wmask && ln1_w && ln1_b && ln2_w && ln2_b && ridge_w && ridge_b && y && ws && stats &&
                  pooled_raw && sumw && zhat && ln2_rstd && z && pred && loss_terms, "head_fwd: null argument");
  VLB_REQUIRE(B > 0 && S > 0 && V > 0 && E % 8 == 0 && E > 0 && E <= 8192, "head_fwd: bad shape B=%d S=%d E=%d V=%d", B, S, E, V);
  hipStream_t st = as_stream(stream);
  const int nblk = vlb_head_partial_rows(S);
  float* partial = ws;
  float* lpart = ws + (int64_t)B * nblk * (E + 2);
  const int ni = (E + 511) / 512;
  int rc;
  if (ni <= 1) rc = launch_pool<1>((const bf16*)hidden, wmask, partial, stats, B, S, E, eps, cu_rows, st);
  else if (ni <= 2) rc = launch_pool<2>((const bf16*)hidden, wmask, partial, stats, B, S, E, eps, cu_rows, st);
  else if (ni <= 4) rc = launch_pool<4>((const bf16*)hidden, wmask, partial, stats, B, S, E, eps, cu_rows, st);
  else if (ni <= 8) rc = launch_pool<8>((const bf16*)hidden, wmask, partial, stats, B, S, E, eps, cu_rows, st);
  else rc = launch_pool<16>((const bf16*)hidden, wmask, partial, stats, B, S, E, eps, cu_rows, st);
  if (rc != VLB_OK) return rc;
  hipLaunchKernelGGL(head_reduce_kernel, dim3((E + 255) / 256, B), dim3(256), 0, st, partial, nblk, pooled_raw, sumw, E);
  VLB_LAUNCH_CHECK();
  hipLaunchKernelGGL(head_ln2_kernel, dim3(B), dim3(1024), 0, st, pooled_raw, sumw, (const bf16*)ln1_w, (const bf16*)ln1_b,
                     (const bf16*)ln2_w, (const bf16*)ln2_b, keep_scale, zhat, ln2_rstd, (bf16*)z, E, eps);
  VLB_LAUNCH_CHECK();
  int rblk = (V + RIDGE_ROWS - 1) / RIDGE_ROWS;
  if (B <= 16 && E % 128 == 0 && E <= 4096) {
    const int ntiles = (V + 15) / 16;
    rblk = ntiles < 2048 ? ntiles : 2048;          // persistent over row tiles; lpart has room for V/16 >= rblk entries
    switch (E / 128) {
#define VLB_RIDGE_CASE(KS) case KS: hipLaunchKernelGGL(ridge_fwd_mfma_kernel<KS>, dim3(rblk), dim3(256), 0, st, (const bf16*)ridge_w, \
                                                       (const bf16*)ridge_b, (const bf16*)z, y, pred, lpart, B, E, V); break;
      VLB_RIDGE_CASE(1) VLB_RIDGE_CASE(2) VLB_RIDGE_CASE(3) VLB_RIDGE_CASE(4) VLB_RIDGE_CASE(5) VLB_RIDGE_CASE(6) VLB_RIDGE_CASE(7)
      VLB_RIDGE_CASE(8) VLB_RIDGE_CASE(9) VLB_RIDGE_CASE(10) VLB_RIDGE_CASE(11) VLB_RIDGE_CASE(12) VLB_RIDGE_CASE(13)
      VLB_RIDGE_CASE(14) VLB_RIDGE_CASE(15) VLB_RIDGE_CASE(16) VLB_RIDGE_CASE(17) VLB_RIDGE_CASE(18) VLB_RIDGE_CASE(19)
      VLB_RIDGE_CASE(20) VLB_RIDGE_CASE(21) VLB_RIDGE_CASE(22) VLB_RIDGE_CASE(23) VLB_RIDGE_CASE(24) VLB_RIDGE_CASE(25)
      VLB_RIDGE_CASE(26) VLB_RIDGE_CASE(27) VLB_RIDGE_CASE(28) VLB_RIDGE_CASE(29) VLB_RIDGE_CASE(30) VLB_RIDGE_CASE(31)
      VLB_RIDGE_CASE(32)
#undef VLB_RIDGE_CASE
    }
  } else {
    const int zlds = (B < BMAX ? B : BMAX) * E * 2;
    if (int rc2 = reserve_ridge_lds(zlds)) return rc2;
    hipLaunchKernelGGL(ridge_fwd_kernel, dim3(rblk), dim3(256), zlds, st, (const bf16*)ridge_w, (const bf16*)ridge_b,
                       (const bf16*)z, y, pred, lpart, B, E, V);
  }
  VLB_LAUNCH_CHECK();
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, lpart, rblk, 1.f / ((float)B * (float)V),
                     l2_lambda, loss_terms);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

extern "C" int vlb_head_bwd(const void* hidden, const float* wmask, const void* ln1_w, const void* ln2_w,
                            const void* ridge_w, const float* y, const float* keep_scale, const float* stats,
                            const float* pooled_raw, const float* sumw, const float* zhat, const float* ln2_rstd,
                            const void* z, const float* pred, float* d_ridge_w, float* d_ridge_b, float* d_ln2_w,
                            float* d_ln2_b, float* d_ln1_w, float* d_ln1_b, float* ws, float* dz_ws, float* dpooled_ws,
                            void* dhidden, int B, int S, int E, int V, float eps, float l2_lambda, float loss_scale,
                            float l2_scale, const int* cu_rows, int total_rows, void* stream) {
  (void)eps;
  VLB_REQUIRE(hidden && wmask && ln1_w && ln2_w && ridge_w && y && stats && pooled_raw && sumw && zhat && ln2_rstd && z &&
                  pred && d_ridge_w && d_ridge_b && d_ln2_w && d_ln2_b && d_ln1_w && d_ln1_b && ws && dz_ws && dpooled_ws,
              "head_bwd: null argument");
  VLB_REQUIRE(B > 0 && S > 0 && V > 0 && E % 8 == 0 && E > 0 && E <= 8192, "head_bwd: bad shape");
  hipStream_t st = as_stream(stream);
  const float gscale = loss_scale * 2.f / ((float)B * (float)V);
  const int rblk = (V + RIDGE_ROWS - 1) / RIDGE_ROWS;
  const int zlds = (B < BMAX ? B : BMAX) * E * 2;
  if (int rc2 = reserve_ridge_lds(zlds)) return rc2;
  hipLaunchKernelGGL(ridge_bwd_w_kernel, dim3(rblk), dim3(256), zlds, st, (const bf16*)ridge_w, (const bf16*)z, pred, y,
                     d_ridge_w, d_ridge_b, B, E, V, gscale, l2_scale * 2.f * l2_lambda);
  VLB_LAUNCH_CHECK();
  if (B <= 16 && E % 8 == 0) {
    // dz[b,e] = sum_v dpred[b,v] W[v,e]: contraction over the ROWS of W -> the MFMA skinny-wgrad kernel
    // (G = dpred^T [V,16] bf16, X = W).  Result rows b < B of a [16,E] fp32 block land in dz_part[0].
    const int splits = vlb_wgrad_splits(V);
    float* wg_ws = ws;                                             // [splits][16][E]
    float* dz16 = ws + (int64_t)splits * 16 * E;                   // [16][E]
    bf16* dpT = reinterpret_cast<bf16*>(dz16 + (int64_t)16 * E);   // [V][16] bf16
    hipLaunchKernelGGL(dpred_t_kernel, dim3((unsigned)(((int64_t)V * 16 + 255) / 256)), dim3(256), 0, st, pred, y, dpT, B, V, gscale);
    VLB_LAUNCH_CHECK();
    int rc3 = vlb_wgrad_skinny(dpT, 16, ridge_w, E, dz16, wg_ws, V, 16, E, 1.f, 0.f, 0.f, nullptr, stream);
    if (rc3 != VLB_OK) return rc3;
    hipLaunchKernelGGL(head_dz_from16_kernel, dim3((E + 255) / 256, B), dim3(256), 0, st, dz16, keep_scale, dz_ws, E);
    VLB_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL(ridge_bwd_z_kernel, dim3((E + 511) / 512, DZ_SPLIT), dim3(256), 0, st, (const bf16*)ridge_w, pred,
                       y, ws, B, E, V, gscale);
    VLB_LAUNCH_CHECK();
    hipLaunchKernelGGL(head_dz_reduce_kernel, dim3((E + 255) / 256, B), dim3(256), 0, st, ws, keep_scale, dz_ws, B, E);
    VLB_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(head_ln2_bwd_kernel, dim3(B), dim3(1024), 0, st, dz_ws, (const bf16*)ln2_w, zhat, ln2_rstd, dpooled_ws, E);
  VLB_LAUNCH_CHECK();
  hipLaunchKernelGGL(head_param_grads_kernel, dim3((E + 255) / 256), dim3(256), 0, st, dz_ws, dpooled_ws, (const bf16*)ln1_w,
                     pooled_raw, sumw, zhat, d_ln2_w, d_ln2_b, d_ln1_w, d_ln1_b, B, E);
  VLB_LAUNCH_CHECK();
  if (dhidden) {
    const int64_t rows = cu_rows ? (int64_t)total_rows : (int64_t)B * S;
    VLB_REQUIRE(rows > 0 && rows <= (int64_t)B * S, "head_bwd: total_rows=%d out of range", total_rows);
    hipLaunchKernelGGL(head_dhidden_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const bf16*)hidden,
                       wmask, stats, dpooled_ws, (bf16*)dhidden, S, E, rows, B, cu_rows);
    VLB_LAUNCH_CHECK();
  }
  return VLB_OK;
}

// ---------------------------------------------------------------- the exported layers on their own
extern "C" int64_t vlb_hrf_pool_ws_floats(int B, int E) { return (int64_t)B * HRF_SPLITS * E; }

extern "C" int vlb_hrf_pool(const void* embeddings, int emb_is_f32, const float* weights, void* out, float* ws, int B, int S, int E,
                            void* stream) {
  VLB_REQUIRE(embeddings && weights && out && ws && B > 0 && S > 0 && E > 0 && E % 8 == 0, "hrf_pool: bad arguments (E must be a multiple of 8)");
  hipStream_t st = as_stream(stream);
  const dim3 grid(HRF_SPLITS, (E + 511) / 512, B), rgrid((E + 255) / 256, B);
  if (emb_is_f32) {
    hipLaunchKernelGGL(hrf_pool_kernel<float>, grid, dim3(256), 0, st, (const float*)embeddings, weights, ws, S, E);
    VLB_LAUNCH_CHECK();
    hipLaunchKernelGGL(hrf_pool_reduce_kernel<float>, rgrid, dim3(256), 0, st, ws, (float*)out, E);
  } else {
    hipLaunchKernelGGL(hrf_pool_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)embeddings, weights, ws, S, E);
    VLB_LAUNCH_CHECK();
    hipLaunchKernelGGL(hrf_pool_reduce_kernel<bf16>, rgrid, dim3(256), 0, st, ws, (bf16*)out, E);
  }
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

extern "C" int64_t vlb_ridge_ws_floats(int V) { return 2 * (int64_t)((V + RIDGE_ROWS - 1) / RIDGE_ROWS) + 8; }

extern "C" int vlb_ridge_fwd(const void* x, const void* ridge_w, const void* ridge_b, float* pred, float* l2_out, float* ws, int B, int E,
                             int V, float l2_lambda, void* stream) {
  VLB_REQUIRE(x && ridge_w && ridge_b && pred && l2_out && ws && B > 0 && V > 0 && E > 0 && E % 8 == 0, "ridge_fwd: bad arguments");
  hipStream_t st = as_stream(stream);
  const int rblk = (V + RIDGE_ROWS - 1) / RIDGE_ROWS;
  const int zlds = (B < BMAX ? B : BMAX) * E * 2;
  if (int rc = reserve_ridge_lds(zlds)) return rc;
  float* terms = ws + 2 * (int64_t)rblk;
  hipLaunchKernelGGL(ridge_fwd_kernel, dim3(rblk), dim3(256), zlds, st, (const bf16*)ridge_w, (const bf16*)ridge_b, (const bf16*)x,
                     (const float*)nullptr, pred, ws, B, E, V);
  VLB_LAUNCH_CHECK();
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, ws, rblk, 1.f / ((float)B * (float)V), l2_lambda, terms);
  VLB_LAUNCH_CHECK();
  hipError_t e = hipMemcpyAsync(l2_out, terms + 1, sizeof(float), hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) { vlb_set_error("ridge_fwd: copy failed: %s", hipGetErrorString(e)); return VLB_ERR_LAUNCH; }
  return VLB_OK;
}
