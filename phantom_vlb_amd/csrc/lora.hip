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
// Same bits when the whole activation has fewer than 2^32 column pairs (every shape of this model): the
// high word of the counter is 0, so its hash `key` = lowbias32(seed) is a per-seed constant (host side)
// and the counter is 32-bit arithmetic.  Kernels pick this path with the uniform flag Seeds::fast.
__device__ __forceinline__ uint32_t drop_bits_fast(uint32_t key, uint32_t row_base, int kpair) {
  return lowbias32((row_base + (uint32_t)kpair) ^ key);
}

struct Seeds { uint32_t s[8]; uint32_t key[8]; uint32_t fast; };

__device__ __forceinline__ uint32_t drop_bits(const Seeds& sd, int g, int64_t m, int K, int kpair) {
  return sd.fast ? drop_bits_fast(sd.key[g], (uint32_t)m * (uint32_t)(K >> 1), kpair) : drop_bits_pair(sd.s[g], m, K, kpair);
}
// keep flags of 8 consecutive columns starting at k0 (k0 % 8 == 0)
__device__ __forceinline__ void keep8(const Seeds& sd, int g, int64_t m, int K, int k0, uint32_t thresh, bool (&keep)[8]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t h = drop_bits(sd, g, m, K, (k0 >> 1) + q);
    keep[2 * q] = (h & 0xffffu) >= thresh;
    keep[2 * q + 1] = (h >> 16) >= thresh;
  }
}

// ---------------------------------------------------------------- t = scale * drop(x) . A^T
// one block = 32 rows of x (two MFMA row tiles share every adapter fragment); its 8 waves each take an
// eighth of K with 4 k-steps of loads in flight, and the partial 16x16 tiles are summed through LDS in a
// fixed order.  MFMA rows = adapter ranks, cols = rows of x.
constexpr int LD_RT = 2;      // row tiles per block
constexpr int LD_WAVES = 8;
template <int G>   // number of 16-rank groups (projections sharing this x)
__global__ __launch_bounds__(64 * LD_WAVES) void lora_down_kernel(const bf16* __restrict__ x, int ldx, const bf16* __restrict__ A,
                                                        bf16* __restrict__ t, int ldt, int M, int K, float scale,
                                                        uint32_t thresh, Seeds seeds) {
  __shared__ float red[LD_WAVES - 1][G][LD_RT][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = blockIdx.x * 16 * LD_RT;
  const int fr = lane & 15, fq = lane >> 4;
  int m[LD_RT];
  const bf16* xr[LD_RT];
#pragma unroll
  for (int r = 0; r < LD_RT; ++r) {
    m[r] = min(m0 + 16 * r + fr, M - 1);
    xr[r] = x + (int64_t)m[r] * ldx + fq * 8;
  }
  f32x4 acc[G][LD_RT];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int r = 0; r < LD_RT; ++r) acc[g][r] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int kq = ((K / 32 + LD_WAVES - 1) / LD_WAVES) * 32;     // K range of this wave, a multiple of 32
  const int kbeg = wave * kq, kend = min(K, kbeg + kq);
  const bf16* ar = A + (int64_t)fr * K + fq * 8;
#pragma unroll 4
  for (int k0 = kbeg; k0 < kend; k0 += 32) {
    bf16x8 xf[LD_RT], af[G];
#pragma unroll
    for (int r = 0; r < LD_RT; ++r) xf[r] = *reinterpret_cast<const bf16x8*>(xr[r] + k0);
#pragma unroll
    for (int g = 0; g < G; ++g) af[g] = *reinterpret_cast<const bf16x8*>(ar + (int64_t)16 * g * K + k0);
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int r = 0; r < LD_RT; ++r) {
        bf16x8 xm = xf[r];
        if (thresh != 0) {
          bool keep[8];
          keep8(seeds, g, m[r], K, k0 + fq * 8, thresh, keep);
#pragma unroll
          for (int j = 0; j < 8; ++j) xm[j] = keep[j] ? xf[r][j] : (bf16)0.f;
        }
        acc[g][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g], xm, acc[g][r], 0, 0, 0);
      }
  }
  if (wave > 0) {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int r = 0; r < LD_RT; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave - 1][g][r][e][lane] = acc[g][r][e];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < LD_RT; ++r) {
      if (m0 + 16 * r + fr >= M) continue;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[g][r][e];
#pragma unroll
          for (int w = 0; w < LD_WAVES - 1; ++w) v += red[w][g][r][e][lane];
          o[e] = (bf16)(v * scale);
        }
        *reinterpret_cast<bf16x4*>(t + (int64_t)(m0 + 16 * r + fr) * ldt + 16 * g + fq * 4) = o;
      }
    }
  }
}

// ---------------------------------------------------------------- dx += keep/(1-p) * (u . A)
// At[K][R] is the transposed adapter (row = input column).  A wave walks 16 rows x 256 columns in steps of
// 64 columns = four MFMAs whose A-operand rows are permuted so that a lane ends up with 16 CONSECUTIVE
// columns of its row (MFMA t, output row i <-> column 16*(i/4) + 4t + i%4): the read-modify-write of dx is
// two 16-byte accesses per lane and the four lanes of a row cover one 128-byte line.
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
  const bool live = m0 + fr < M;
  // B operand: u[m][16g + 8fq .. +8] for fq < 2, zeros for the padded half of the k=32 step
  bf16x8 uf[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    uf[g] = bf16x8{};
    if (fq < 2) uf[g] = *reinterpret_cast<const bf16x8*>(u + (int64_t)m * ldu + 16 * g + 8 * fq);
  }
  const int arow = 16 * (fr >> 2) + (fr & 3);          // + 4t: At row (relative to kcol) feeding MFMA t, output row fr
  bf16* prow = dx + (int64_t)m * lddx + 16 * fq;
  bf16x8 old0 = *reinterpret_cast<const bf16x8*>(prow + kbase);
  bf16x8 old1 = *reinterpret_cast<const bf16x8*>(prow + kbase + 8);
  for (int kc = 0; kc < 256 && kbase + kc < K; kc += 64) {
    const int kcol = kbase + kc;
    bf16x8 nx0 = old0, nx1 = old1;
    if (kc + 64 < 256 && kcol + 64 < K) {               // next step's dx is in flight during this step's math
      nx0 = *reinterpret_cast<const bf16x8*>(prow + kcol + 64);
      nx1 = *reinterpret_cast<const bf16x8*>(prow + kcol + 72);
    }
    float sum[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) sum[i] = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        bf16x8 af = bf16x8{};
        if (fq < 2) af = *reinterpret_cast<const bf16x8*>(At + (int64_t)(kcol + arow + 4 * t) * ldat + 16 * g + 8 * fq);
        f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, uf[g], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        // lane holds columns kcol + 16fq + 4t + {0..3} of row m
        if (thresh != 0) {
          const int kk = kcol + 16 * fq + 4 * t;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const uint32_t h = drop_bits(seeds, g, m, K, (kk >> 1) + q);
            if ((h & 0xffffu) < thresh) d[2 * q] = 0.f;
            if ((h >> 16) < thresh) d[2 * q + 1] = 0.f;
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) sum[4 * t + e] += d[e] * inv_keep;
      }
    }
    if (live) {
      bf16x8 o0, o1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o0[e] = (bf16)((float)old0[e] + sum[e]);
        o1[e] = (bf16)((float)old1[e] + sum[8 + e]);
      }
      *reinterpret_cast<bf16x8*>(prow + kcol) = o0;
      *reinterpret_cast<bf16x8*>(prow + kcol + 8) = o1;
    }
    old0 = nx0; old1 = nx1;
  }
}

// ---------------------------------------------------------------- skinny wgrad (MFMA)
// dW[16G, K] partials: contraction over the ROWS m of G[m, 16G] and X[m, K].  Both MFMA operands are
// therefore "k-strided": the X tile [32 rows][256 cols] and the G tile [32 rows][16G] are staged
// row-major in LDS (X by LDS-DMA, swizzled on the source) and read with ds_read_b64_tr_b16.
// Block = 4 waves, 256 columns of K; wave = 64 columns (4 MFMA column tiles) x 16G ranks.
// grid (ceil(K/256), splits); ws[split][16G][K] fp32 partial slabs, summed in a fixed order afterwards.
constexpr int WG_COLS = 256;
constexpr int WG_STEP = 32;     // rows of M per MFMA k-step

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16_l(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                   (void __attribute__((address_space(3)))*)l, 16, 0, 0);
}
// 32-byte pair slot swizzle of the X tile: rows r and r+8 and the 4 rows of a transposed read all differ
__device__ __forceinline__ int xsw(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

// WITH_U (G = 1): the same pass over X also forms the block's slice of  u = X . Bt^T  (contraction over this
// block's 256 columns): wave w multiplies its 64 columns of every 32-row tile by the matching 64 columns of
// Bt [16, K] (row reads of the X image, MFMA rows = ranks), the four waves' partials meet in LDS and are stored as
// upart[column block][row][16] fp32 - summed over column blocks by u_reduce_kernel.  This is LoRA backward's
// u = s dy B riding on the dB^T = t^T dy pass, instead of a second sweep over dy.
// WITH_U over several projections that share one dy = X (q|k|v, gate|up): column block c0 belongs to projection
// j = #{col0[1..] <= c0}; its G tile is t[:, 16j..16j+16), its Bt is UProj::bt[j] ([16, n_j], n_j = col0[j+1] - col0[j]).
// The slab / upart layouts do not change (they are indexed by the column of X), only the reduce kernel scatters.
struct UProj { int col0[4]; const bf16* bt[3]; float* dW[3]; int n; };

template <int G, bool WITH_U = false>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(const bf16* __restrict__ Gm, int ldg, const bf16* __restrict__ X,
                                                         int ldx, float* __restrict__ ws, int M, int K,
                                                         int rows_per_split, uint32_t thresh, Seeds seeds,
                                                         UProj up = UProj{}, float* __restrict__ upart = nullptr) {
  constexpr int XT = WG_STEP * WG_COLS * 2;          // 16 KB
  constexpr int GT = WG_STEP * 16 * G * 2;           // 1 KB per group
  constexpr int UT = WITH_U ? 2 * 4 * 2 * 4 * 64 * 4 : 0;     // [parity][wave][row tile][reg][lane] fp32 = 16 KB
  __shared__ __attribute__((aligned(16))) char smem[2 * (XT + GT) + UT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c0 = blockIdx.x * WG_COLS;
  const int r0 = blockIdx.y * rows_per_split, r1 = min(M, r0 + rows_per_split);
  const int fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
  const int N = 16 * G;

  f32x4 acc[G][4];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[g][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // u = X . Bt^T: this wave's 64 columns = two k-steps of 32; Bt fragments stay in registers for the whole block
  bf16x8 btf[2];
  if constexpr (WITH_U) {
    const int pj = (c0 >= up.col0[1]) + (c0 >= up.col0[2]);            // block-uniform projection index
    const int pc0 = up.col0[pj], pn = up.col0[pj + 1] - pc0;
    const bf16* Bt = up.bt[pj];
    Gm += 16 * pj;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int lc = c0 - pc0 + wave * 64 + 32 * ks + 8 * fq;          // column inside the projection
      btf[ks] = *reinterpret_cast<const bf16x8*>(Bt + (int64_t)fr * pn + min(lc, pn - 8));
      if (lc >= pn) btf[ks] = bf16x8{};
    }
  }
  float* ured = reinterpret_cast<float*>(smem + 2 * (XT + GT));

  // staging: X piece i of this wave = rows 2*(4i+wave) + (lane>>5); 16-byte position pc = lane & 31.  The G tile (32 rows x
  // 16G columns, at most one 16-byte chunk per thread) goes through registers: its load is ISSUED in front of the X pieces
  // and written to LDS only after the current step's math (g_store) - a wait for it in between would also wait for the
  // LDS-DMA pieces behind it (vmcnt counts in order) and serialise the next tile's latency with this tile's compute.
  const int g_r = tid / (2 * G), g_ch = tid % (2 * G);
  const bool g_mine = tid < WG_STEP * 2 * G;
  auto stage = [&](int buf, int m0, bf16x8& gv) {
    char* xb = smem + buf * (XT + GT);
    gv = bf16x8{};                               // rows beyond the split are zero: they contribute nothing
    if (g_mine && m0 + g_r < r1) gv = *reinterpret_cast<const bf16x8*>(Gm + (int64_t)(m0 + g_r) * ldg + g_ch * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = 4 * i + wave, r = 2 * piece + (lane >> 5), pc = lane & 31;
      const int chunk = (((pc >> 1) ^ xsw(r)) << 1) | (pc & 1);
      const int m = min(m0 + r, M - 1);                        // rows past the end are zeroed through G
      const int col = min(c0 + chunk * 8, K - 8);
      glds16_l(X + (int64_t)m * ldx + col, xb + piece * 1024);
    }
  };
  auto g_store = [&](int buf, const bf16x8& gv) {
    if (g_mine) *reinterpret_cast<bf16x8*>(smem + buf * (XT + GT) + XT + g_r * 32 * G + g_ch * 16) = gv;
  };

  const int nsteps = (r1 - r0 + WG_STEP - 1) / WG_STEP;
  bf16x8 gnext;
  if (nsteps > 0) { stage(0, r0, gnext); g_store(0, gnext); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    const int cur = st & 1, m0 = r0 + st * WG_STEP;
    const char* xb = smem + cur * (XT + GT);
    const char* gb = xb + XT;
    // ---- 1. every fragment of the current tile into registers (the compiler orders LDS accesses behind in-flight LDS-DMA
    // with vmcnt(0), so nothing may touch LDS between the issue of the next tile and the end of this step's math)
    // A' fragments (G^T): rows 8fq + {0..3} and + 4, columns 16g + 4tp
    bf16x8 af[G], xf[4];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(gb + (8 * fq + tq) * 32 * G + (16 * g + 4 * tp) * 2));
      const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(gb + (8 * fq + 4 + tq) * 32 * G + (16 * g + 4 * tp) * 2));
      af[g] = __builtin_bit_cast(bf16x8, (s16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int colb = wave * 64 + n * 16;                  // column block inside the tile
      // B' fragment (X): lane fr gets column colb+fr of rows 8fq + {0..7}
      const int ra = 8 * fq + tq, rb = ra + 4;
      const int pa = (((colb >> 4) ^ xsw(ra)) << 5) + ((4 * tp) << 1);
      const int pb = (((colb >> 4) ^ xsw(rb)) << 5) + ((4 * tp) << 1);
      const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(xb + ra * 512 + pa));
      const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(xb + rb * 512 + pb));
      xf[n] = __builtin_bit_cast(bf16x8, (s16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
    bf16x8 xr[2][2];
    if constexpr (WITH_U) {
      // row reads of the X image: source chunk c (8 columns) of row r sits at 16-byte position
      // (((c>>1) ^ xsw(r)) << 1) | (c&1) of the row's 512 bytes
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int r = 16 * t + fr, c = wave * 8 + 4 * ks + fq;       // 16-byte source chunk inside the 256-column tile
          const int pc = (((c >> 1) ^ xsw(r)) << 1) | (c & 1);
          xr[t][ks] = *reinterpret_cast<const bf16x8*>(xb + r * 512 + pc * 16);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- 2. the next tile goes in flight under this tile's math
    if (st + 1 < nsteps) stage(cur ^ 1, m0 + WG_STEP, gnext);
    __builtin_amdgcn_sched_barrier(0);
    // ---- 3. math
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int col = c0 + wave * 64 + n * 16 + fr;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        bf16x8 xm = xf[n];
        if (thresh != 0) {
          // lanes fr and fr^1 hold the two columns of one hashed pair: the even lane hashes rows 0..3, the
          // odd lane rows 4..7, and they swap results (quad_perm [1,0,3,2]) - half the hashes per lane
          const bool odd = col & 1;
          uint32_t hm[4], ho[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            hm[jj] = drop_bits(seeds, g, m0 + 8 * fq + (odd ? 4 : 0) + jj, K, col >> 1);
            ho[jj] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hm[jj], 0xB1, 0xf, 0xf, true);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const uint32_t hsh = (j < 4) == !odd ? hm[j & 3] : ho[j & 3];
            const uint32_t bits = odd ? (hsh >> 16) : (hsh & 0xffffu);
            if (bits < thresh) xm[j] = (bf16)0.f;
          }
        }
        acc[g][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g], xm, acc[g][n], 0, 0, 0);
      }
    }
    if constexpr (WITH_U) {
      float* mine = ured + ((cur * 4 + wave) * 2) * 4 * 64;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 ua = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) ua = __builtin_amdgcn_mfma_f32_16x16x32_bf16(btf[ks], xr[t][ks], ua, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) mine[(t * 4 + e) * 64 + lane] = ua[e];
      }
    }
    if (st + 1 < nsteps) g_store(cur ^ 1, gnext);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (WITH_U) {
      // sum the four waves' partials of this step (fixed order) and store [row][16 ranks]: 512 values, 2 per thread
      const float* base = ured + cur * 4 * 2 * 4 * 64;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int idx = tid + 256 * q;                         // (t, e, lane)
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) v += base[w * 2 * 4 * 64 + idx];
        const int l = idx & 63, e = (idx >> 6) & 3, t = idx >> 8;
        const int row = m0 + 16 * t + (l & 15), rank = 4 * (l >> 4) + e;
        if (row < r1) upart[((int64_t)blockIdx.x * M + row) * 16 + rank] = v;
      }
    }
  }
  // partial slab: lane owns column c0 + wave*64 + n*16 + fr of ranks 16g + 4fq + {0..3}
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int col = c0 + wave * 64 + n * 16 + fr;
      if (col < K) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ws[((int64_t)blockIdx.y * N + 16 * g + 4 * fq + e) * K + col] = acc[g][n][e];
      }
    }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dW, int splits, int64_t nk, float alpha,
                                    float beta) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nk; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < splits; ++k) s += ws[(int64_t)k * nk + i];
    dW[i] = alpha * s + (beta != 0.f ? beta * dW[i] : 0.f);
  }
}

// ---------------------------------------------------------------- derived adapter layouts
// Every derived layout is "a [16, n] bf16 matrix written transposed into 16 columns of an [n, ld] image":
// A_j [16,K] -> A^T padded (At[:, 16j:16j+16]) and B_j^T [16,N] -> the block-diagonal padded B
// (Bpad[row0:row0+N, 16j:16j+16]).  One launch walks a device table of 256-row jobs.
// dst already at (row0, col0).  il = 1: destination rows follow the gate/up interleave of VLB_ACT_SWIGLU_PAIR - source
// column i lands in row (i / 16) * 32 + i % 16 (dst pre-offset by 16 rows for the up projection).
struct ScatterJob { const bf16* src; bf16* dst; int n; int n0; int ld; int il; };

__global__ __launch_bounds__(256) void transpose16_scatter_kernel(const ScatterJob* __restrict__ jobs) {
  const ScatterJob j = jobs[blockIdx.x];
  const int i = j.n0 + threadIdx.x;
  if (i >= j.n) return;
  bf16x8 lo, hi;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    lo[r] = j.src[(int64_t)r * j.n + i];
    hi[r] = j.src[(int64_t)(r + 8) * j.n + i];
  }
  bf16* d = j.dst + (int64_t)(j.il ? ((i >> 4) * 32 + (i & 15)) : i) * j.ld;
  *reinterpret_cast<bf16x8*>(d) = lo;
  *reinterpret_cast<bf16x8*>(d + 8) = hi;
}

// One launch for both fixed-order sums of the fused dB/u pass: blocks [0, wblocks) reduce the dW slabs (grid-stride) and
// scatter them to the projections' [16, n_j] gradients, the rest compute u[m][16j + r] (bf16, row stride ldu) =
// scale * sum over projection j's column blocks of upart[block][m][r].
__global__ void wgrad_u_reduce_kernel(const float* __restrict__ ws, int splits, int K, float alpha, float beta, int wblocks,
                                      const float* __restrict__ upart, bf16* __restrict__ u, int ldu, int M, float scale, UProj up) {
  const int64_t nk = (int64_t)16 * K;
  if ((int)blockIdx.x < wblocks) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nk; i += (int64_t)wblocks * blockDim.x) {
      float s = 0.f;
#pragma unroll 8
      for (int k = 0; k < splits; ++k) s += ws[(int64_t)k * nk + i];          // loads independent: 8 in flight
      const int r = (int)(i / K), col = (int)(i - (int64_t)r * K);
      const int pj = (col >= up.col0[1]) + (col >= up.col0[2]);
      float* d = up.dW[pj] + (int64_t)r * (up.col0[pj + 1] - up.col0[pj]) + (col - up.col0[pj]);
      *d = alpha * s + (beta != 0.f ? beta * *d : 0.f);
    }
    return;
  }
  const int w = 16 * up.n;
  const int64_t i = (int64_t)(blockIdx.x - wblocks) * blockDim.x + threadIdx.x;
  if (i >= (int64_t)M * w) return;
  const int m = (int)(i / w), jr = (int)(i - (int64_t)m * w), pj = jr >> 4, r = jr & 15;
  const int b0 = up.col0[pj] / WG_COLS, b1 = (up.col0[pj + 1] + WG_COLS - 1) / WG_COLS;      // (one projection: any K; several: multiples of 256)
  float v = 0.f;
#pragma unroll 8
  for (int b = b0; b < b1; ++b) v += upart[((int64_t)b * M + m) * 16 + r];
  u[(int64_t)m * ldu + jr] = (bf16)(v * scale);
}

inline uint32_t lowbias32_host(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
inline Seeds make_seeds(const uint32_t* seeds_host, int groups, int M, int K) {
  Seeds sd{};
  for (int g = 0; g < groups; ++g) {
    sd.s[g] = seeds_host ? seeds_host[g] : 0u;
    sd.key[g] = lowbias32_host(sd.s[g]);
  }
  sd.fast = ((int64_t)M * (K >> 1) < (1ll << 32)) ? 1u : 0u;
  return sd;
}
inline uint32_t thresh16(float p) {
  if (p <= 0.f) return 0;
  uint32_t t = (uint32_t)(p * 65536.f + 0.5f);
  return t > 65535u ? 65535u : t;
}
}  // namespace

// Row splits of the skinny wgrad sweeps (grid = K/256 column blocks x splits).  Small M: >= 256 rows (8 MFMA k-steps) per split.
// From 4096 rows on: 32 splits (>= 128 rows each), so that the grid is K/8 workgroups - a whole number of workgroups per CU for
// every K that is a multiple of 2048 (4096: 2 per CU, 6144: 3, 14336: 7, 28672: 14).  With ceil(M/256) = 23 splits at the LoRA batch
// the K = 4096 sweeps ran 368 workgroups on 256 CUs (112 CUs with two, 144 with one), and the hash-bound ones took as long as
// their doubly loaded CUs.
extern "C" int vlb_wgrad_splits(int M) {
  if (M >= 4096) return 32;
  int s = (M + 255) / 256;
  return s < 1 ? 1 : s;
}

extern "C" int vlb_wgrad_skinny(const void* G, int ldg, const void* X, int ldx, float* dW, float* ws, int M, int N, int K,
                                float alpha, float beta, float drop_p, const uint32_t* seeds_host, void* stream) {
  VLB_REQUIRE(G && X && dW && ws, "wgrad_skinny: null operand");
  VLB_REQUIRE(M > 0 && (N == 16 || N == 32 || N == 48) && K >= 8 && K % 8 == 0 && ldx % 8 == 0 && ldg % 8 == 0,
              "wgrad_skinny: bad shape M=%d N=%d K=%d (N must be 16, 32 or 48)", M, N, K);
  VLB_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seeds_host), "wgrad_skinny: bad dropout arguments");
  VLB_REQUIRE((((uintptr_t)G | (uintptr_t)X) % 16) == 0, "wgrad_skinny: operands must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  const int splits = vlb_wgrad_splits(M);
  const int rps = (((M + splits - 1) / splits) + WG_STEP - 1) / WG_STEP * WG_STEP;
  const float inv_keep = 1.f / (1.f - drop_p);
  const Seeds sd = make_seeds(seeds_host, N / 16, M, K);
  dim3 grid((K + WG_COLS - 1) / WG_COLS, splits);
  const uint32_t th = thresh16(drop_p);
  switch (N / 16) {
    case 1: hipLaunchKernelGGL(wgrad_mfma_kernel<1>, grid, dim3(256), 0, st, (const bf16*)G, ldg, (const bf16*)X, ldx, ws, M, K, rps, th, sd); break;
    case 2: hipLaunchKernelGGL(wgrad_mfma_kernel<2>, grid, dim3(256), 0, st, (const bf16*)G, ldg, (const bf16*)X, ldx, ws, M, K, rps, th, sd); break;
    default: hipLaunchKernelGGL(wgrad_mfma_kernel<3>, grid, dim3(256), 0, st, (const bf16*)G, ldg, (const bf16*)X, ldx, ws, M, K, rps, th, sd); break;
  }
  VLB_LAUNCH_CHECK();
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
  const Seeds s = make_seeds(seeds_host, R / 16, M, K);
  const float sc = scale / (1.f - drop_p);
  dim3 grid((M + 16 * LD_RT - 1) / (16 * LD_RT));
  hipStream_t st = as_stream(stream);
  const uint32_t th = thresh16(drop_p);
  switch (R / 16) {
    case 1: hipLaunchKernelGGL(lora_down_kernel<1>, grid, dim3(64 * LD_WAVES), 0, st, (const bf16*)x, ldx, (const bf16*)A, (bf16*)t, ldt, M, K, sc, th, s); break;
    case 2: hipLaunchKernelGGL(lora_down_kernel<2>, grid, dim3(64 * LD_WAVES), 0, st, (const bf16*)x, ldx, (const bf16*)A, (bf16*)t, ldt, M, K, sc, th, s); break;
    default: hipLaunchKernelGGL(lora_down_kernel<3>, grid, dim3(64 * LD_WAVES), 0, st, (const bf16*)x, ldx, (const bf16*)A, (bf16*)t, ldt, M, K, sc, th, s); break;
  }
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

extern "C" int vlb_lora_dx_masked(const void* u, int ldu, const void* At, int ldat, void* dx, int lddx, int M, int K, int R,
                                  float drop_p, const uint32_t* seeds_host, void* stream) {
  VLB_REQUIRE(u && At && dx, "lora_dx_masked: null operand");
  VLB_REQUIRE(M > 0 && K % 64 == 0 && R % 16 == 0 && R >= 16 && R <= 48 && ldu % 8 == 0 && lddx % 8 == 0 && ldat % 8 == 0 && ldat >= R &&
                  ((uintptr_t)dx % 16) == 0,
              "lora_dx_masked: bad shape M=%d K=%d R=%d", M, K, R);
  VLB_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seeds_host), "lora_dx_masked: bad dropout arguments");
  const Seeds s = make_seeds(seeds_host, R / 16, M, K);
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

extern "C" int vlb_transpose16_scatter(const void* jobs, int n_jobs, void* stream) {
  VLB_REQUIRE(jobs && n_jobs > 0, "transpose16_scatter: empty job table");
  hipLaunchKernelGGL(transpose16_scatter_kernel, dim3(n_jobs), dim3(256), 0, as_stream(stream), (const ScatterJob*)jobs);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

extern "C" int64_t vlb_wgrad_u_ws_floats(int M, int K) { return (int64_t)((K + WG_COLS - 1) / WG_COLS) * M * 16; }

static int wgrad_skinny_u_impl(const void* G, int ldg, const void* X, int ldx, int M, int K, int nproj, const int* cols,
                               float* const* dW, const void* const* Bt, float* ws, float alpha, float beta, float u_scale, void* u,
                               int ldu, float* u_ws, void* stream) {
  UProj up{};
  up.n = nproj;
  int c = 0;
  for (int j = 0; j < 4; ++j) up.col0[j] = 0x7fffffff;
  for (int j = 0; j < nproj; ++j) { up.col0[j] = c; up.bt[j] = (const bf16*)Bt[j]; up.dW[j] = dW[j]; c += cols[j]; }
  up.col0[nproj] = c;
  hipStream_t st = as_stream(stream);
  const int splits = vlb_wgrad_splits(M);
  const int rps = (((M + splits - 1) / splits) + WG_STEP - 1) / WG_STEP * WG_STEP;
  const Seeds sd = make_seeds(nullptr, 1, M, K);
  dim3 grid((K + WG_COLS - 1) / WG_COLS, splits);
  hipLaunchKernelGGL((wgrad_mfma_kernel<1, true>), grid, dim3(256), 0, st, (const bf16*)G, ldg, (const bf16*)X, ldx, ws, M, K, rps,
                     0u, sd, up, u_ws);
  VLB_LAUNCH_CHECK();
  const int64_t nk = (int64_t)16 * K;
  int wblocks = (int)((nk + 255) / 256); if (wblocks > 1024) wblocks = 1024;
  const int ublocks = (int)(((int64_t)M * 16 * nproj + 255) / 256);
  hipLaunchKernelGGL(wgrad_u_reduce_kernel, dim3(wblocks + ublocks), dim3(256), 0, st, ws, splits, K, alpha, beta, wblocks,
                     u_ws, (bf16*)u, ldu, M, u_scale, up);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

extern "C" int vlb_wgrad_skinny_u(const void* G, int ldg, const void* X, int ldx, float* dW, float* ws, int M, int K,
                                  float alpha, float beta, const void* Bt, float u_scale, void* u, int ldu, float* u_ws,
                                  void* stream) {
  VLB_REQUIRE(G && X && dW && ws && Bt && u && u_ws, "wgrad_skinny_u: null operand");
  VLB_REQUIRE(M > 0 && K >= 64 && K % 64 == 0 && ldx % 8 == 0 && ldg % 8 == 0 && ldu >= 16,
              "wgrad_skinny_u: bad shape M=%d K=%d", M, K);
  VLB_REQUIRE((((uintptr_t)G | (uintptr_t)X | (uintptr_t)Bt) % 16) == 0, "wgrad_skinny_u: operands must be 16-byte aligned");
  return wgrad_skinny_u_impl(G, ldg, X, ldx, M, K, 1, &K, &dW, &Bt, ws, alpha, beta, u_scale, u, ldu, u_ws, stream);
}

extern "C" int vlb_wgrad_skinny_u_multi(const void* G, int ldg, const void* X, int ldx, int M, int nproj, const int* cols,
                                        float* const* dW, const void* const* Bt, float* ws, float alpha, float beta, float u_scale,
                                        void* u, int ldu, float* u_ws, void* stream) {
  VLB_REQUIRE(G && X && cols && dW && Bt && ws && u && u_ws && nproj >= 1 && nproj <= 3, "wgrad_skinny_u_multi: bad arguments (1..3 projections)");
  int K = 0;
  for (int j = 0; j < nproj; ++j) {
    VLB_REQUIRE(dW[j] && Bt[j] && cols[j] > 0 && cols[j] % WG_COLS == 0 && ((uintptr_t)Bt[j] % 16) == 0,
                "wgrad_skinny_u_multi: projection %d: columns must be a multiple of %d, Bt 16-byte aligned", j, WG_COLS);
    K += cols[j];
  }
  VLB_REQUIRE(M > 0 && ldx % 8 == 0 && ldx >= K && ldg % 8 == 0 && ldg >= 16 * nproj && ldu >= 16 * nproj, "wgrad_skinny_u_multi: bad leading dimensions");
  VLB_REQUIRE((((uintptr_t)G | (uintptr_t)X) % 16) == 0, "wgrad_skinny_u_multi: operands must be 16-byte aligned");
  return wgrad_skinny_u_impl(G, ldg, X, ldx, M, K, nproj, cols, dW, Bt, ws, alpha, beta, u_scale, u, ldu, u_ws, stream);
}
