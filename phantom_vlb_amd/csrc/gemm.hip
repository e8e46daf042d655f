// bf16 MFMA GEMMs for gfx950:  C[M,N] = act(A[M,K].W[N,K]^T + A2.W2^T + bias) + residual
//
// Two kernels:
//  * gemm_tile_kernel<BM,BN,WM,WN>: 512 threads (8 waves), BK=64, LDS-DMA staging
//    (global_load_lds_dwordx4, swizzle applied on the SOURCE address), double-buffered LDS,
//    v_mfma_f32_16x16x32_bf16, XCD-aware tile order.  Needs N % BN == 0, K % 64 == 0; any M.
//  * gemm_generic_kernel: 64x64 tile, bounds-checked register staging, any M/N, K % 8 == 0.
//
// MFMA operand roles are swapped on purpose (rows of the MFMA = output COLUMNS n, columns of the
// MFMA = output rows m): each lane then owns 4 consecutive n of one row m per 16x16 tile, so the
// epilogue packs 4 bf16 into one 8-byte store and bias/residual are 8-byte loads.
#include <atomic>
#include <type_traits>

#include "common.hpp"

namespace {

constexpr int BK = 64;           // K elements per LDS tile row (128 bytes)
constexpr int ROW_BYTES = BK * 2;

struct GemmArgs {
  const bf16* A;  const bf16* W;  bf16* C;
  const bf16* A2; const bf16* W2;
  const bf16* bias; const bf16* residual;
  int M, N, K, K2;
  int lda, ldw, ldc, ldr, lda2, ldw2;
  int act;
  int tiles_m, tiles_n;     // grid of PARENT tiles (BM x BN*split_n) the tile order is defined on
  int tile0;                // first parent tile of this launch (the launch covers [tile0, tile0 + grid/split_n))
  int split_n;              // a workgroup computes 1/split_n of a parent tile's columns (tail launches)
  int grid;                 // workgroups of this launch (host side only)
  // masked second pair (LoRA backward through dropout): C = A.W^T + keep(m,n)/(1-p) * (A2.W2^T), keep from the
  // counter hash of lora.hip (fast 32-bit path): bits(m, n) = lowbias32((m*(N/2) + n/2) ^ key), 16 bits per column
  uint32_t drop_thresh, drop_key; float drop_scale;
  // split-K tail launch: the tiles of a partial last wave are each cut into k_splits contiguous K ranges, one
  // workgroup per (split, tile); raw fp32 accumulators go to ws[split][tile][wave][i][j][lane][4] and
  // gemm_splitk_reduce_kernel sums the splits in a fixed order and applies the epilogue
  float* ws; int k_splits, tail_tiles;
  int order;                // tile order (see tile_coords / xcd_remap_pid)
  // SWIGLU_PAIR only: when set, the pre-activations are ALSO stored, de-interleaved, as [gate | up] rows of N columns
  // (what a LoRA / full backward needs) - vlb_gemm_swiglu_save
  bf16* aux; int ldaux;
  int stagger;              // tools build only (timing experiment): odd workgroups of the first round start this many 10-ns ticks late
  int wide;                 // C (and aux) rows 16-byte aligned: required by the four-wave kernels, which store 16 bytes per lane (store_pair16)
  int persist_iters;        // tools build only (persistent experiment): output tiles per workgroup, grid = 256
  // stream-K launch (gemm_w4_kernel<.., STREAMK>): 256 workgroups share sk_total = tiles x K-tiles iterations evenly; see the kernel
  int sk_total; unsigned long long* sk_flags; unsigned long long sk_want; int* sk_err;
};

// blockIdx -> (m0, n0).  The order is defined on the full parent grid, so a GEMM can be cut into several launches (full
// waves with 256x256 tiles + a partial last wave re-tiled 256x128 or split along K) at any tile index.  Product order
// (GemmArgs::order = 3): column bands of 8 tiles with the rows walked inside a band, so 32 consecutive tiles are 4 rows x
// 8 columns (the 32 workgroups resident on one XCD share 4 A + 8 W panels in its L2), and xcd_remap_pid deals every
// round of 256 tiles to the eight XCDs in chunks of 32: at any moment the whole chip sweeps the row panels of A against
// the SAME 8 W panels, so a W panel comes out of HBM once and A (48-77 MB here) is re-read from the Infinity Cache.
// Against contiguous per-XCD runs over bands of 4 row tiles (order 0: 8 XCDs in 8 different places, 96 distinct panels
// per round instead of ~40) the same kernel measured +2 % (LoRA batch) / +7 % (frozen batch) on gate/up, -1.5 % / -2.3 %
// on the whole step (tools/ab_step_variant.py).  Orders 0-7 exist for A/B in the tools build.  When A is the larger operand
// and too large for the Infinity Cache the host picks order 2 instead (row bands dealt across the XCDs; pick_order).
__device__ __forceinline__ void tile_coords(const GemmArgs& p, int pid, int sub, int BM, int BN, int& m0, int& n0) {
  int tm, tn;
  if (p.order & 4) {
    // super-bands of 16 columns walked in groups of 4 rows, each group as two 4 x 8 chunks: with the chunks of a round
    // dealt to the XCDs (order bit 1) a chip-wide round is a 16 x 16 block of tiles = 16 A + 16 W panels
    const int sbt = 16 * p.tiles_m;
    const int sb = pid / sbt, c0 = sb * 16;
    const int wd = min(p.tiles_n - c0, 16);
    int w = pid - sb * sbt;
    const int rg = w / (4 * wd);
    const int gsz = min(p.tiles_m - 4 * rg, 4);
    w -= rg * 4 * wd;
    const int h = w / (gsz * 8);
    const int cw = min(wd - 8 * h, 8);
    w -= h * gsz * 8;
    tm = 4 * rg + w / cw;
    tn = c0 + 8 * h + w % cw;
  } else if (p.order & 1) {
    // column bands of 8 tiles, rows walked inside a band: 32 consecutive tiles = 4 rows x 8 columns, and one chip-wide
    // round of 256 tiles sweeps (almost) every row panel of A against the same 8 W panels
    constexpr int GROUP_N = 8;
    const int band = GROUP_N * p.tiles_m;
    const int cb = pid / band, c0 = cb * GROUP_N;
    const int csz = min(p.tiles_n - c0, GROUP_N);
    const int w = pid - cb * band;
    tm = w / csz;
    tn = c0 + w % csz;
  } else {
    constexpr int GROUP_M = 4;
    const int band = GROUP_M * p.tiles_n;
    const int g0 = (pid / band) * GROUP_M;
    const int gsz = min(p.tiles_m - g0, GROUP_M);
    tm = g0 + (pid % band) % gsz;
    tn = (pid % band) / gsz;
  }
  m0 = tm * BM;
  n0 = (tn * p.split_n + sub) * BN;
}

// blocks b and b+8 run on the same XCD.  order bit 1 clear: each XCD gets one contiguous run of the launch's work items;
// set: every round of 256 work items is dealt out in chunks of 32 (XCD x takes items [32x, 32x+32) of the round), so the
// eight XCDs work on neighbouring tiles at the same time and share what they pull through the Infinity Cache.
__device__ __forceinline__ int xcd_remap_pid(int order) {
  const int nblk = gridDim.x, pid = blockIdx.x;
  const int x = pid & 7, j = pid >> 3;
  int base = 0, n = nblk, jj = j;
  if (order & 2) {
    const int T = nblk >> 8, t = j >> 5;
    if (t < T) return (t << 8) + (x << 5) + (j & 31);
    base = T << 8; n = nblk - base; jj = j - (T << 5);
  }
  const int q = n >> 3, r = n & 7;
  return base + (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + jj;
}

// returns the K split of this workgroup (0 unless the launch is a split-K tail)
__device__ __forceinline__ int map_tile(const GemmArgs& p, int BM, int BN, int& m0, int& n0) {
  int pid = xcd_remap_pid(p.order);
  int sub = 0, ks = 0;
  if (p.split_n > 1) { sub = pid % p.split_n; pid /= p.split_n; }
  if (p.k_splits > 1) { ks = pid / p.tail_tiles; pid -= ks * p.tail_tiles; }   // split-major: an XCD's run shares one K range
  tile_coords(p, pid + p.tile0, sub, BM, BN, m0, n0);
  return ks;
}

__device__ __forceinline__ float apply_act(float x, int act) {
  switch (act) {
    case VLB_ACT_QUICK_GELU: return quick_gelu_f(x);
    case VLB_ACT_GELU: return gelu_erf_f(x);
    case VLB_ACT_SILU: return silu_f(x);
    default: return x;
  }
}

// (store_pair16, the 16-byte epilogue store shared with gemm_fp8.hip, lives in common.hpp)
// SWIGLU_PAIR epilogue piece: g / u = four consecutive gate / up pre-activations of row m (fragments 2j and 2j+1 of the
// interleaved weight), n = their column in the N/2-wide output.  With p.aux the pre-activations are kept too.
__device__ __forceinline__ void store_swiglu4(const GemmArgs& p, const f32x4& g, const f32x4& u, int m, int n) {
  bf16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (bf16)(silu_f(g[e]) * u[e]);
  *reinterpret_cast<bf16x4*>(p.C + (int64_t)m * p.ldc + n) = o;
  if (p.aux) {
    bf16x4 gb, ub;
#pragma unroll
    for (int e = 0; e < 4; ++e) { gb[e] = (bf16)g[e]; ub[e] = (bf16)u[e]; }
    bf16* ap = p.aux + (int64_t)m * p.ldaux + n;
    *reinterpret_cast<bf16x4*>(ap) = gb;
    *reinterpret_cast<bf16x4*>(ap + (p.N >> 1)) = ub;
  }
}

// two adjacent output fragments of the SWIGLU_PAIR epilogue (accumulator fragments 4q..4q+3) with 16-byte stores
__device__ __forceinline__ void store_swiglu8(const GemmArgs& p, const f32x4& g0, const f32x4& u0, const f32x4& g1, const f32x4& u1,
                                              int m, int n, int fq) {
  bf16x4 h0, h1;
#pragma unroll
  for (int e = 0; e < 4; ++e) { h0[e] = (bf16)(silu_f(g0[e]) * u0[e]); h1[e] = (bf16)(silu_f(g1[e]) * u1[e]); }
  store_pair16(p.C + (int64_t)m * p.ldc, n, h0, h1, fq);
  if (p.aux) {
    bf16x4 ga, gb, ua, ub;
#pragma unroll
    for (int e = 0; e < 4; ++e) { ga[e] = (bf16)g0[e]; gb[e] = (bf16)g1[e]; ua[e] = (bf16)u0[e]; ub[e] = (bf16)u1[e]; }
    bf16* ap = p.aux + (int64_t)m * p.ldaux;
    store_pair16(ap, n, ga, gb, fq);
    store_pair16(ap + (p.N >> 1), n, ua, ub, fq);
  }
}

// bias / activation / residual of one fragment piece: lane holds C[m][n .. n+3]
// ACT is a compile-time parameter: every fully unrolled epilogue nest then holds the code of ONE activation (see the
// instruction-cache note at w4_store_frag2); bias and residual stay run-time tests (a load and an add each).
template <int ACT>
__device__ __forceinline__ float act_apply(float x) {
  if constexpr (ACT == VLB_ACT_QUICK_GELU) return quick_gelu_f(x);
  else if constexpr (ACT == VLB_ACT_GELU) return gelu_erf_f(x);
  else if constexpr (ACT == VLB_ACT_SILU) return silu_f(x);
  else return x;
}
template <int ACT>
__device__ __forceinline__ bf16x4 w4_frag_value(const GemmArgs& p, f32x4 v, int m, int n) {
  if (p.bias) {
    const bf16x4 bb = *reinterpret_cast<const bf16x4*>(p.bias + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += (float)bb[e];
  }
  if constexpr (ACT != VLB_ACT_NONE) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act_apply<ACT>(v[e]);
  }
  if (p.residual) {
    const bf16x4 rr = *reinterpret_cast<const bf16x4*>(p.residual + (int64_t)m * p.ldr + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += (float)rr[e];
  }
  bf16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
  return o;
}
// run `BODY(ACT)` with the launch's activation as a compile-time constant (uniform dispatch, one nest executes)
#define VLB_DISPATCH_ACT(act, BODY)                        \
  do {                                                     \
    if ((act) == VLB_ACT_QUICK_GELU) { BODY(VLB_ACT_QUICK_GELU) } \
    else if ((act) == VLB_ACT_GELU) { BODY(VLB_ACT_GELU) }        \
    else if ((act) == VLB_ACT_SILU) { BODY(VLB_ACT_SILU) }        \
    else { BODY(VLB_ACT_NONE) }                                   \
  } while (0)

// LDS image of a [rows][64] bf16 tile: 128-byte rows, 16-byte chunk c of row r lives at slot
// c ^ ((r>>1)&7).  Two rows share one 256-byte bank row, so the 16 rows x 1 chunk column that a
// ds_read_b128 lane group touches land on 16 distinct 16-byte slots: conflict-free.
__device__ __forceinline__ int lds_off(int r, int c) { return r * ROW_BYTES + ((c ^ ((r >> 1) & 7)) << 4); }

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                   (void __attribute__((address_space(3)))*)l, 16, 0, 0);
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(512, 2) void gemm_tile_kernel(GemmArgs p) {
  static_assert(WM * WN == 8, "8 waves");
  constexpr int TM = BM / WM, TN = BN / WN;   // per-wave output tile
  constexpr int MT = TM / 16, NT = TN / 16;   // 16x16 MFMA tiles per wave
  constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int A_LD = BM / 64, B_LD = BN / 64;  // glds instructions per thread per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int m0, n0;
  map_tile(p, BM, BN, m0, n0);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- staging addresses: instruction i of this wave fills LDS rows [i*64 + wave*8, +8)
  const int srow = (lane >> 3), sslot = lane & 7;
  const bf16* a_src[A_LD]; const bf16* b_src[B_LD];
  const bf16* a2_src[A_LD]; const bf16* b2_src[B_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int r = i * 64 + wave * 8 + srow;
    const int c = sslot ^ ((r >> 1) & 7);
    const int gm = min(m0 + r, p.M - 1);
    a_src[i] = p.A + (int64_t)gm * p.lda + c * 8;
    a2_src[i] = p.A2 ? p.A2 + (int64_t)gm * p.lda2 + c * 8 : nullptr;
  }
#pragma unroll
  for (int i = 0; i < B_LD; ++i) {
    const int r = i * 64 + wave * 8 + srow;
    const int c = sslot ^ ((r >> 1) & 7);
    b_src[i] = p.W + (int64_t)(n0 + r) * p.ldw + c * 8;
    b2_src[i] = p.W2 ? p.W2 + (int64_t)(n0 + r) * p.ldw2 + c * 8 : nullptr;
  }
  const int nk1 = p.K / BK;
  const int nk = nk1 + p.K2 / BK;

  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE + wave * 8 * ROW_BYTES;
    if (kt < nk1) {
      const int ko = kt * BK;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) glds16(a_src[i] + ko, base + i * 64 * ROW_BYTES);
#pragma unroll
      for (int i = 0; i < B_LD; ++i) glds16(b_src[i] + ko, base + A_BYTES + i * 64 * ROW_BYTES);
    } else {
      const int ko = (kt - nk1) * BK;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) glds16(a2_src[i] + ko, base + i * 64 * ROW_BYTES);
#pragma unroll
      for (int i = 0; i < B_LD; ++i) glds16(b2_src[i] + ko, base + A_BYTES + i * 64 * ROW_BYTES);
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (within a stage): row = tile_row0 + (lane&15), chunk = ks*4 + (lane>>4)
  const int fr = lane & 15, fq = lane >> 4;
  int a_off[MT][2], b_off[NT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a_off[i][ks] = lds_off(wm * TM + i * 16 + fr, ks * 4 + fq);
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) b_off[j][ks] = A_BYTES + lds_off(wn * TN + j * 16 + fr, ks * 4 + fq);

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* sb = smem + cur * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[MT], wf[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sb + b_off[j][ks]);
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sb + a_off[i][ks]);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: lane owns n = nbase + fq*4 + {0..3} of row m = mbase + fr, per 16x16 tile
  // C rows 16-byte aligned (p.wide): adjacent fragments paired into 16-byte stores (store_pair16); one nest per activation
#define PP_EPILOGUE_WIDE(ACT)                                                                                            \
  _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                                       \
    const int m = m0 + wm * TM + i * 16 + fr;                                                                            \
    if (m >= p.M) continue;                                                                                              \
    _Pragma("unroll") for (int j = 0; j < NT; j += 2) {                                                                  \
      const int n = n0 + wn * TN + j * 16 + fq * 4;                                                                      \
      store_pair16(p.C + (int64_t)m * p.ldc, n, w4_frag_value<ACT>(p, acc[i][j], m, n), w4_frag_value<ACT>(p, acc[i][j + 1], m, n + 16), fq); \
    }                                                                                                                    \
  }
#define PP_EPILOGUE_NARROW(ACT)                                                                                          \
  _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                                       \
    const int m = m0 + wm * TM + i * 16 + fr;                                                                            \
    if (m >= p.M) continue;                                                                                              \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                                     \
      const int n = n0 + wn * TN + j * 16 + fq * 4;                                                                      \
      *reinterpret_cast<bf16x4*>(p.C + (int64_t)m * p.ldc + n) = w4_frag_value<ACT>(p, acc[i][j], m, n);                 \
    }                                                                                                                    \
  }
  if (p.wide) VLB_DISPATCH_ACT(p.act, PP_EPILOGUE_WIDE);
  else VLB_DISPATCH_ACT(p.act, PP_EPILOGUE_NARROW);
#undef PP_EPILOGUE_WIDE
#undef PP_EPILOGUE_NARROW
}

// ------------------------------------------------------------------------------------------------
// Ping-pong variant of the tile kernel.  The 8 waves form two groups of 4 (one wave of each group per
// SIMD).  Every K-tile is two phases (k-steps of 32); a phase is a LOAD segment (12 ds_read_b128 of
// fragments, plus the LDS-DMA issue of the tile after next) and a COMPUTE segment (32 MFMAs), separated
// by workgroup barriers.  Group 1 runs one barrier interval behind group 0, so on every SIMD one wave
// is always in its MFMA segment while its partner reads LDS: the matrix pipe never waits for loads.
//   interval:      0    1    2    3    4   ...
//   group 0:       L0   C0   L1   C1   L2  ...
//   group 1:       -    L0   C0   L1   C1  ...
// LDS-DMA for tile t+1 is issued in the first LOAD segment of tile t (its buffer was last read two
// intervals earlier, and every LOAD segment ends with lgkmcnt(0) before its barrier), and is waited for
// with ONE vmcnt(0) per tile placed just before the barrier that precedes the first read of tile t+1,
// i.e. after ~3-4 intervals of flight.
// ------------------------------------------------------------------------------------------------
// S0..S3: how many of the tile's LDS-DMA instructions a wave issues in load slot 0..3 of the 4-interval
// window that precedes the tile's first read (slot q of group 0 = its segments L0,C0,L1,C1 of tile t;
// of group 1 = C1 of tile t-1, then L0,C0,L1 of tile t).  S0+S1+S2+S3 = A_LD + B_LD.
// ABL (timing-only ablations, results are WRONG when non-zero): bit0 skip LDS-DMA in the loop, bit1 skip
// fragment reads in the loop, bit2 skip the barriers in the loop.
template <int BM, int BN, int WM, int WN, int S0, int S1, int S2, int S3, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(GemmArgs p) {
  static_assert(WM * WN == 8 && WM % 2 == 0, "8 waves, groups split along M");
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MT = TM / 16, NT = TN / 16;
  constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int A_LD = BM / 64, B_LD = BN / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int m0, n0;
  map_tile(p, BM, BN, m0, n0);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int grp = wave >> 2;                       // 0: leads, 1: trails by one barrier interval

  const int srow = (lane >> 3), sslot = lane & 7;
  const bf16* a_src[A_LD]; const bf16* b_src[B_LD];
  const bf16* a2_src[A_LD]; const bf16* b2_src[B_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int r = i * 64 + wave * 8 + srow;
    const int c = sslot ^ ((r >> 1) & 7);
    const int gm = min(m0 + r, p.M - 1);
    a_src[i] = p.A + (int64_t)gm * p.lda + c * 8;
    a2_src[i] = p.A2 ? p.A2 + (int64_t)gm * p.lda2 + c * 8 : nullptr;
  }
#pragma unroll
  for (int i = 0; i < B_LD; ++i) {
    const int r = i * 64 + wave * 8 + srow;
    const int c = sslot ^ ((r >> 1) & 7);
    b_src[i] = p.W + (int64_t)(n0 + r) * p.ldw + c * 8;
    b2_src[i] = p.W2 ? p.W2 + (int64_t)(n0 + r) * p.ldw2 + c * 8 : nullptr;
  }
  const int nk1 = p.K / BK;
  const int nk = nk1 + p.K2 / BK;

  static_assert(S0 + S1 + S2 + S3 == A_LD + B_LD, "slot counts must cover the tile");
  // instruction index i of a tile: 0..A_LD-1 = A pieces, A_LD.. = W pieces
  auto stage_range = [&](int buf, int kt, int from, int to) {
    char* base = smem + buf * STAGE + wave * 8 * ROW_BYTES;
    const bool first = kt < nk1;
    const int ko = (first ? kt : kt - nk1) * BK;
#pragma unroll
    for (int i = 0; i < A_LD + B_LD; ++i) {
      if (i < from || i >= to) continue;
      if (i < A_LD) glds16((first ? a_src[i] : a2_src[i]) + ko, base + i * 64 * ROW_BYTES);
      else glds16((first ? b_src[i - A_LD] : b2_src[i - A_LD]) + ko, base + A_BYTES + (i - A_LD) * 64 * ROW_BYTES);
    }
  };
  auto stage = [&](int buf, int kt) { stage_range(buf, kt, 0, A_LD + B_LD); };
  constexpr int O1 = S0, O2 = S0 + S1, O3 = S0 + S1 + S2, O4 = A_LD + B_LD;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  int a_off[MT][2], b_off[NT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a_off[i][ks] = lds_off(wm * TM + i * 16 + fr, ks * 4 + fq);
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) b_off[j][ks] = A_BYTES + lds_off(wn * TN + j * 16 + fr, ks * 4 + fq);

#define PP_BARRIER()                                  \
  do {                                                \
    __builtin_amdgcn_sched_barrier(0);                \
    __builtin_amdgcn_s_barrier();                     \
    __builtin_amdgcn_sched_barrier(0);                \
  } while (0)
#define PP_LOOP_BARRIER()                             \
  do {                                                \
    if constexpr (!(ABL & 4)) { PP_BARRIER(); } else { __builtin_amdgcn_sched_barrier(0); } \
  } while (0)
#define PP_LGKM0() __builtin_amdgcn_s_waitcnt(0xc07f)
#define PP_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

  stage(0, 0);
  if (nk > 1) stage(1, 1);
  PP_VM0();
  PP_BARRIER();
  if (grp == 1) PP_BARRIER();

  bf16x8 af[MT], wf[NT];
  for (int kt = 0; kt < nk; ++kt) {
    const char* sb = smem + (kt & 1) * STAGE;
    const int nb = (kt & 1) ^ 1;                       // buffer of tile kt+1 (and of tile kt+2: kt&1)
    const bool ld1 = !(ABL & 1) && kt >= 1 && kt + 1 < nk;   // this window loads tile kt+1
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // ---------------- LOAD segment
#pragma unroll
      for (int j = 0; j < NT; ++j) if ((ABL & 2) == 0 || kt == 0) wf[j] = *reinterpret_cast<const bf16x8*>(sb + b_off[j][ks]);
#pragma unroll
      for (int i = 0; i < MT; ++i) if ((ABL & 2) == 0 || kt == 0) af[i] = *reinterpret_cast<const bf16x8*>(sb + a_off[i][ks]);
      if (ld1) {
        if (grp == 0) { if (ks == 0) stage_range(nb, kt + 1, 0, O1); else stage_range(nb, kt + 1, O2, O3); }
        else          { if (ks == 0) stage_range(nb, kt + 1, O1, O2); else stage_range(nb, kt + 1, O3, O4); }
      }
      PP_LGKM0();
      if (ks == 1 && grp == 1 && kt >= 1) PP_VM0();
      PP_LOOP_BARRIER();
      // ---------------- COMPUTE segment
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      if (grp == 0) {
        if (ld1) { if (ks == 0) stage_range(nb, kt + 1, O1, O2); else stage_range(nb, kt + 1, O3, O4); }
      } else {
        if (ks == 0) { if (ld1) stage_range(nb, kt + 1, O2, O3); }
        else if (!(ABL & 1) && kt + 2 < nk) stage_range(kt & 1, kt + 2, 0, O1);   // slot 0 of the NEXT window
      }
      if (ks == 1 && grp == 0 && kt >= 1) PP_VM0();
      PP_LOOP_BARRIER();
    }
  }
  if (grp == 0) PP_BARRIER();
#undef PP_BARRIER
#undef PP_LOOP_BARRIER
#undef PP_LGKM0
#undef PP_VM0

  if (p.act == VLB_ACT_SWIGLU_PAIR) {
    // W rows are interleaved in 16-row blocks [gate_b | up_b]: n-tiles (2j, 2j+1) hold gate and up of the
    // same 16 output features at the same lane/register positions -> out = silu(gate) * up, N/2 columns.
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + wm * TM + i * 16 + fr;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NT; j += 2) {
        store_swiglu4(p, acc[i][j], acc[i][j + 1], m, (n0 + wn * TN) / 2 + (j / 2) * 16 + fq * 4);
      }
    }
    return;
  }
  // C rows 16-byte aligned (p.wide): adjacent fragments paired into 16-byte stores (store_pair16); one nest per activation
#define PP_EPILOGUE_WIDE(ACT)                                                                                            \
  _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                                       \
    const int m = m0 + wm * TM + i * 16 + fr;                                                                            \
    if (m >= p.M) continue;                                                                                              \
    _Pragma("unroll") for (int j = 0; j < NT; j += 2) {                                                                  \
      const int n = n0 + wn * TN + j * 16 + fq * 4;                                                                      \
      store_pair16(p.C + (int64_t)m * p.ldc, n, w4_frag_value<ACT>(p, acc[i][j], m, n), w4_frag_value<ACT>(p, acc[i][j + 1], m, n + 16), fq); \
    }                                                                                                                    \
  }
#define PP_EPILOGUE_NARROW(ACT)                                                                                          \
  _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                                       \
    const int m = m0 + wm * TM + i * 16 + fr;                                                                            \
    if (m >= p.M) continue;                                                                                              \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                                     \
      const int n = n0 + wn * TN + j * 16 + fq * 4;                                                                      \
      *reinterpret_cast<bf16x4*>(p.C + (int64_t)m * p.ldc + n) = w4_frag_value<ACT>(p, acc[i][j], m, n);                 \
    }                                                                                                                    \
  }
  if (p.wide) VLB_DISPATCH_ACT(p.act, PP_EPILOGUE_WIDE);
  else VLB_DISPATCH_ACT(p.act, PP_EPILOGUE_NARROW);
#undef PP_EPILOGUE_WIDE
#undef PP_EPILOGUE_NARROW
}

// ------------------------------------------------------------------------------------------------
// generic bounds-checked kernel: 64x64 tile, 256 threads (2x2 waves, each 32x32 = 2x2 MFMA tiles),
// BK = 32, register staging into padded LDS rows.  Correct for every shape with K % 8 == 0.
// ------------------------------------------------------------------------------------------------
constexpr int GK = 32;
constexpr int GLD = GK + 8;   // padded row (80 bytes): 16-byte aligned, breaks power-of-two stride

__global__ __launch_bounds__(256) void gemm_generic_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) bf16 sA[64 * GLD];
  __shared__ __attribute__((aligned(16))) bf16 sW[64 * GLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int fr = lane & 15, fq = lane >> 4;
  // each thread stages one 16-byte chunk of A and one of W per k-tile: row = tid/4, chunk = tid%4
  const int lr = tid >> 2, lc = tid & 3;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int pass = 0; pass < 2; ++pass) {
    const bf16* A = pass ? p.A2 : p.A;
    const bf16* W = pass ? p.W2 : p.W;
    const int K = pass ? p.K2 : p.K;
    const int lda = pass ? p.lda2 : p.lda, ldw = pass ? p.ldw2 : p.ldw;
    if (K == 0 || A == nullptr) continue;
    for (int k0 = 0; k0 < K; k0 += GK) {
      bf16x8 va = {}, vw = {};
      const int kk = k0 + lc * 8;
      if (m0 + lr < p.M && kk < K) va = *reinterpret_cast<const bf16x8*>(A + (int64_t)(m0 + lr) * lda + kk);
      if (n0 + lr < p.N && kk < K) vw = *reinterpret_cast<const bf16x8*>(W + (int64_t)(n0 + lr) * ldw + kk);
      __syncthreads();
      *reinterpret_cast<bf16x8*>(&sA[lr * GLD + lc * 8]) = va;
      *reinterpret_cast<bf16x8*>(&sW[lr * GLD + lc * 8]) = vw;
      __syncthreads();
      bf16x8 af[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(&sA[(wm * 32 + i * 16 + fr) * GLD + fq * 8]);
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(&sW[(wn * 32 + j * 16 + fr) * GLD + fq * 8]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
  }
  if (p.act == VLB_ACT_SWIGLU_PAIR) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + wm * 32 + i * 16 + fr;
      if (m >= p.M) continue;
      const int nb = (n0 + wn * 32) / 2 + fq * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (2 * (nb + e) < p.N) {
          p.C[(int64_t)m * p.ldc + nb + e] = (bf16)(silu_f(acc[i][0][e]) * acc[i][1][e]);
          if (p.aux) {
            p.aux[(int64_t)m * p.ldaux + nb + e] = (bf16)acc[i][0][e];
            p.aux[(int64_t)m * p.ldaux + (p.N >> 1) + nb + e] = (bf16)acc[i][1][e];
          }
        }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 32 + i * 16 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nb = n0 + wn * 32 + j * 16 + fq * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = nb + e;
        if (n >= p.N) continue;
        float v = acc[i][j][e];
        if (p.bias) v += (float)p.bias[n];
        v = apply_act(v, p.act);
        if (p.residual) v += (float)p.residual[(int64_t)m * p.ldr + n];
        p.C[(int64_t)m * p.ldc + n] = (bf16)v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Four-wave variant of the 256x256x64 tile: 256 threads = 2x2 waves, each wave owns a 128x128 block
// (8x8 MFMA tiles, 256 accumulator registers - one wave per SIMD with the full 512-register budget).
// Why: per K-tile the 8-wave kernels read (128+64)*64*2 B * 8 = 192 KB of fragments from LDS plus the
// 64 KB LDS-DMA write = 2048 LDS cycles at 128 B/clk - exactly the 2048 MFMA cycles of the tile, so the
// two pipes must overlap perfectly.  128x128 wave blocks read (128+128)*64*2 B * 4 = 128 KB (1536 cycles
// with the DMA): a quarter of the LDS time becomes slack, and one barrier per K-tile replaces four.
// Schedule of one wave (software pipelined, no partner wave to hide behind):
//   block 1:  MFMAs of k-step 0   || ds_reads of k-step 1 fragments
//   lgkmcnt(0), vmcnt(0) (tile t+1 landed; it was issued 1.5 tiles ago), s_barrier
//   block 2:  MFMAs of k-step 1   || LDS-DMA of tile t+2 into the buffer just released
//                                 || ds_reads of tile t+1's k-step 0 fragments
// Instruction mix per 4 MFMAs (64 cycles of matrix pipe): one ds_read_b128 (+ one LDS-DMA in block 2).
// ------------------------------------------------------------------------------------------------
// MFMA with the accumulator tied in place in an AGPR tuple.  With all 256 AGPRs holding accumulators the
// register allocator otherwise rotates destination tuples through copies; the asm form pins dst == srcC.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma_tied(f32x4& c, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
               : "+a"(c)
               : "v"(__builtin_bit_cast(i32x4_t, a)), "v"(__builtin_bit_cast(i32x4_t, b)));
}

// epilogue arithmetic of one 16x16 accumulator fragment: lane holds C[m][n .. n+3]
// SWIGLU_BWD: v = d(silu(gate)*up) for columns n..n+3 of a [M, N = ff] product; residual = the saved [gate | up] activations
// (row stride ldr), C = [d gate | d up] (row stride ldc): SwiGLU backward without materialising v
__device__ __forceinline__ void w4_swiglu_bwd_math(const f32x4& v, const bf16x4& g4, const bf16x4& u4, bf16x4& dg, bf16x4& du) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float g = (float)g4[e], u = (float)u4[e];
    const float sg = sigmoid_f(g);
    du[e] = (bf16)(v[e] * (g * sg));
    dg[e] = (bf16)(v[e] * u * (sg * (1.f + g * (1.f - sg))));
  }
}
// Two adjacent fragments (columns n.. and n+16..), 16-byte stores.  KIND picks ONE epilogue at compile time so that each
// kind's fully unrolled loop nest is a compact, contiguous instruction stream: with the kinds as run-time branches inside
// every fragment the executed path hopped over ~280 KB of never-executed activation code per tile - instruction-cache
// misses that cost the plain epilogue ~8 % of a whole K = 4096 GEMM (1.11 -> 1.03 ms on gate/up at the LoRA batch).
enum { EPI_PLAIN = 0, EPI_RESIDUAL = 1, EPI_SWIGLU_BWD = 2, EPI_GENERIC = 3 };     // EPI_GENERIC + act (0..3): bias / activation / residual
__device__ __forceinline__ bf16x4 cvt4(const f32x4& v) {
  bf16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
  return o;
}
template <int KIND>
__device__ __forceinline__ void w4_store_frag2(const GemmArgs& p, const f32x4& va, const f32x4& vb, int m, int n, int fq) {
  bf16* crow = p.C + (int64_t)m * p.ldc;
  if constexpr (KIND == EPI_SWIGLU_BWD) {
    // saved [gate | up] rows: one 16-byte load each for this lane's two fragment pieces (rows 16-byte aligned: host check)
    const bf16* grow = p.residual + (int64_t)m * p.ldr;
    bf16x4 ga, gb, ua, ub, dga, dua, dgb, dub;
    load_pair16(grow, n, fq, ga, gb);
    load_pair16(grow + p.N, n, fq, ua, ub);
    w4_swiglu_bwd_math(va, ga, ua, dga, dua);
    w4_swiglu_bwd_math(vb, gb, ub, dgb, dub);
    store_pair16(crow, n, dga, dgb, fq);
    store_pair16(crow + p.N, n, dua, dub, fq);
  } else if constexpr (KIND == EPI_PLAIN) {
    store_pair16(crow, n, cvt4(va), cvt4(vb), fq);
  } else if constexpr (KIND == EPI_RESIDUAL) {
    const bf16* rp = p.residual + (int64_t)m * p.ldr + n;
    const bf16x4 ra = *reinterpret_cast<const bf16x4*>(rp), rb = *reinterpret_cast<const bf16x4*>(rp + 16);
    f32x4 xa = va, xb = vb;
#pragma unroll
    for (int e = 0; e < 4; ++e) { xa[e] += (float)ra[e]; xb[e] += (float)rb[e]; }
    store_pair16(crow, n, cvt4(xa), cvt4(xb), fq);
  } else {       // EPI_GENERIC + ACT: KIND = EPI_GENERIC + the activation code
    constexpr int ACT = KIND - EPI_GENERIC;
    store_pair16(crow, n, w4_frag_value<ACT>(p, va, m, n), w4_frag_value<ACT>(p, vb, m, n + 16), fq);
  }
}
// which epilogue a launch needs (uniform): bias or an activation -> generic
__device__ __forceinline__ int w4_epilogue_kind(const GemmArgs& p) {
  if (p.act == VLB_ACT_SWIGLU_BWD) return EPI_SWIGLU_BWD;
  if (p.bias || p.act != VLB_ACT_NONE) return EPI_GENERIC;
  return p.residual ? EPI_RESIDUAL : EPI_PLAIN;
}
// NT = 8: 256x256 tile (wave block 128x128).  NT = 4: 256x128 tile (wave block 128x64) for the re-cut tiles
// of a partial last wave - same pipeline, 8 MFMA groups per block instead of 16.  MT = 6: 192-row tiles (wave
// block 96 rows), picked by the host when they quantise the row count into fewer, fuller waves of tiles.
// MASKED: the second operand pair (one K-tile, K2 = 64) runs FIRST, the accumulators are then multiplied by the
// dropout keep mask / (1-p) in place, and the main K loop continues on top - dx = dy.W + keep*(u.A)/(1-p) in one GEMM.
//
// STREAMK (instantiated by the tools build only; measured 1.4-2x slower than the round + tail plans, see launch_w4_streamk):
// ONE launch of 256 workgroups for an output of more than 256 tiles that is not a whole number of rounds.  The
// launch's work is the sequence of (tile, K-tile) iterations in tile order, tiles x nk of them; workgroup r (= the r-th CU
// slot of its XCD-contiguous run) takes iterations [bound(r), bound(r+1)) - the same number for everybody, so nobody idles
// through a mostly empty last round and no second launch re-cuts it.  A range is at least one tile long (tiles >= 256), so a
// tile is shared by at most two workgroups: the one that holds its FIRST K-tiles (at the END of its own range) owns it, the
// next workgroup computes the rest of the tile at the START of its range, stores the raw fp32 accumulators as a slab
// (layout of the split-K tail) and raises the tile's flag; the owner adds the slab to its own accumulators (fixed order:
// bit-reproducible) and runs the epilogue.  Whole tiles inside a range end in the normal epilogue.  A contributor never
// waits for anything, and an owner only for the first segment of the workgroup dispatched right after it in the same XCD
// run, so the launch cannot deadlock however few CUs the dispatcher finds free; the owner's wait is bounded all the same
// (sk_err is raised and the host fails the next call).  Cross-XCD visibility: slab and flag travel with agent-scope (sc1)
// stores and loads - the instructions the gfx942 / gfx950 memory model uses for agent-scope atomics - ordered by vmcnt(0) +
// a workgroup barrier on the writing side and by the flag load + barrier on the reading side; no cache-wide write-back or
// invalidate (see the contributor branch).
template <int NT = 8, int ABL = 0, int MT = 8, bool MASKED = false, bool SPLITK = false>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(GemmArgs p) {
  constexpr bool STREAMK = (ABL & 256) != 0;           // (an ABL bit, not a template parameter of its own: the product kernels keep their names)
  constexpr int TM = 16 * MT, TN = 16 * NT, BM = 2 * TM, BN = 2 * TN;
  constexpr int NG = MT * NT / 4;                      // groups of 4 MFMAs per block (one k-step of the wave block)
  constexpr int LAST_A = NG / 2 + 1;                   // group in which the double-slotted last A fragment is fetched
  constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int A_LD = BM / 32, B_LD = BN / 32;       // LDS-DMA instructions per thread per tile (32 rows each)
  constexpr int ND = A_LD + B_LD;
  static_assert((NT == 8 || NT == 4) && (MT == 8 || MT == 6), "wave block 128|96 rows x 128|64 columns");
  constexpr bool ROWSPLIT = (ABL & 64) != 0;            // the K loop that splits a K-tile's MFMAs by output rows instead of by k-step (below)
  static_assert(ROWSPLIT || (ND <= 2 * NG && (NT - 1) < NG && (2 * (MT - 2) + 2) * NT / 8 < NG), "schedule does not fit the block");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  constexpr bool PERSIST = (ABL & 128) != 0;             // tools: one workgroup per CU streams over persist_iters output tiles (below)
  static_assert(!PERSIST || (!MASKED && !SPLITK && !ROWSPLIT), "persistent stream: plain launches only");
  static_assert(!STREAMK || (NT == 8 && !SPLITK && !PERSIST && !ROWSPLIT), "stream-K: whole-width tiles, its own K ranges");
  constexpr bool KRANGE = SPLITK || STREAMK;           // this workgroup walks a sub-range of a tile's K-tiles
  constexpr int SK_MIN = 4;                             // no stream-K segment shorter than this many K-tiles
  int m0, n0;
  int ksplit = 0;
  const int nk1 = MASKED ? p.K2 / BK : p.K / BK;       // K-tiles of the pair that runs first
  const int nk_all = p.K / BK + p.K2 / BK;
  // stream-K: iteration range of this workgroup.  Boundaries closer than SK_MIN K-tiles to a tile boundary snap onto it
  // (both neighbours compute the same bound).  XCD x = blockIdx & 7 takes the contiguous ranges [32x, 32x + 32).
  [[maybe_unused]] int sk_r = 0, sk_it = 0, sk_end = 0;
  auto sk_bound = [&](int r) {
    int b = (int)((int64_t)r * p.sk_total / 256);
    const int rem = b % nk_all;
    if (rem < SK_MIN) b -= rem; else if (rem > nk_all - SK_MIN) b += nk_all - rem;
    return b;
  };
  // persistent stream: work item `it` of workgroup b is tile id it*256 + (b&7)*32 + (b>>3) - the tile a launch of
  // persist_iters*256 workgroups hands to the same CU slot in its round `it` (tile order bit 1)
  auto persist_coords = [&](int it, int& pm0, int& pn0) {
    tile_coords(p, p.tile0 + it * 256 + ((blockIdx.x & 7) << 5) + (blockIdx.x >> 3), 0, BM, BN, pm0, pn0);
  };
  if constexpr (PERSIST) persist_coords(0, m0, n0);
  else if constexpr (STREAMK) {
    sk_r = ((blockIdx.x & 7) << 5) + (blockIdx.x >> 3);
    sk_it = sk_bound(sk_r); sk_end = sk_bound(sk_r + 1);
    if (sk_it >= sk_end) return;
    tile_coords(p, p.tile0 + sk_it / nk_all, 0, BM, BN, m0, n0);
  }
  else ksplit = map_tile(p, BM, BN, m0, n0);
#ifdef VLB_TOOLS
  // timing experiment: de-synchronise the CUs (every workgroup of a launch otherwise starts, and reaches its epilogue's
  // burst of memory traffic, at the same moment): odd workgroups of the first round of 256 wait p.stagger x 10 ns
  if (p.stagger > 0 && blockIdx.x < 256 && (blockIdx.x & 8)) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)p.stagger) __builtin_amdgcn_s_sleep(32);
  }
#endif

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // staging: instruction i of this wave fills LDS rows [i*32 + wave*8, +8) of A (i < A_LD) or W.
  // Sources are (uniform base) + 32-bit byte offset, recomputed per instruction from two row numbers:
  // no per-instruction pointer arrays (the 128 fragment + 256 accumulator registers leave no room).
  // The swizzle term ((r>>1)&7) is the same for rows r and r + 32*i.
  const int srow = (lane >> 3), sslot = lane & 7;
  const int r0 = wave * 8 + srow;
  const int colb = (sslot ^ ((r0 >> 1) & 7)) * 8;
  // (tools build, timing only: ABL bit 4 / 5 make every workgroup stage W / A panel 0 - operands that always hit in L2)
  int rowA = (ABL & 32) ? r0 : m0 + r0, rowW = (ABL & 16) ? r0 : n0 + r0;       // (re-pointed per output tile by the persistent stream / per stream-K segment)
  // split-K tail: this workgroup walks K-tiles [kt_lo, kt_lo + nk) of the concatenated (pair 1 | pair 2) sequence
  int kt_lo = SPLITK ? (int)((int64_t)ksplit * nk_all / p.k_splits) : 0;
  int nk = SPLITK ? (int)((int64_t)(ksplit + 1) * nk_all / p.k_splits) - kt_lo : nk_all;
  if constexpr (STREAMK) { kt_lo = sk_it % nk_all; nk = min(nk_all - kt_lo, sk_end - sk_it); }     // first segment (re-set per segment below)

  // LDS-DMA sources = uniform base (SGPR pair, advanced 128 B per K-tile) + a per-instruction 32-bit
  // byte offset held in a VGPR: the loop issues each global_load_lds with no address arithmetic at all.
  // The offsets are rebuilt once when the K loop crosses into the second operand pair (A2, W2).
  uint32_t offA[A_LD], offW[B_LD];
  const char* curA; const char* curW;
  auto set_operands = [&](const bf16* A_, const bf16* W_, int lda_, int ldw_, int kt_in_pair) {
#pragma unroll
    for (int i = 0; i < A_LD; ++i)
      offA[i] = ((uint32_t)min(rowA + 32 * i, p.M - 1) * (uint32_t)lda_ + (uint32_t)colb) * 2u;
#pragma unroll
    for (int j = 0; j < B_LD; ++j) offW[j] = ((uint32_t)(rowW + 32 * j) * (uint32_t)ldw_ + (uint32_t)colb) * 2u;
    curA = reinterpret_cast<const char*>(A_) + (int64_t)kt_in_pair * ROW_BYTES;
    curW = reinterpret_cast<const char*>(W_) + (int64_t)kt_in_pair * ROW_BYTES;
  };
  auto select = [&](int rel) {          // call with consecutive rel = 0, 1, ..: positions the bases on K-tile kt_lo + rel
    const int kt = KRANGE ? rel + kt_lo : rel;
    if (rel == 0) {
      const bool second = KRANGE && kt >= nk1;
      if (MASKED != second) set_operands(p.A2, p.W2, p.lda2, p.ldw2, second ? kt - nk1 : kt);
      else set_operands(p.A, p.W, p.lda, p.ldw, second ? kt - nk1 : kt);
    }
    else if (kt == nk1) { if (MASKED) set_operands(p.A, p.W, p.lda, p.ldw, 0); else set_operands(p.A2, p.W2, p.lda2, p.ldw2, 0); }
    else { curA += ROW_BYTES; curW += ROW_BYTES; }
  };
  // persistent stream: K-tile kt2 = kt + 2 of the CURRENT output tile, which for kt2 >= nk is K-tile kt2 - nk of the NEXT
  // one (its coordinates in nm0 / nn0; the host guarantees nk >= 4 and K2 == 0 or K2 >= BK)
  int nm0 = 0, nn0 = 0, p_it = 0, p_par = 0;
  auto select_p = [&](int kt2) __attribute__((always_inline)) {
    if (kt2 == nk) {
      persist_coords(p_it + 1, nm0, nn0);
      rowA = nm0 + r0; rowW = nn0 + r0;
      set_operands(p.A, p.W, p.lda, p.ldw, 0);
    } else if (kt2 == nk1 && kt2 < nk) { set_operands(p.A2, p.W2, p.lda2, p.ldw2, 0); }
    else { curA += ROW_BYTES; curW += ROW_BYTES; }
  };
  auto dma = [&](int buf, int i) {     // instruction i of the selected tile (0..A_LD-1: A, then W)
    char* base = smem + buf * STAGE + wave * 8 * ROW_BYTES;
    if (i < A_LD) glds16(curA + offA[i], base + i * 32 * ROW_BYTES);
    else glds16(curW + offW[i - A_LD], base + A_BYTES + (i - A_LD) * 32 * ROW_BYTES);
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment offsets: tile i adds i*16 rows = i*2048 bytes (the swizzle term does not depend on i)
  const int fr = lane & 15, fq = lane >> 4;
  int a_off[2], b_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_off[ks] = lds_off(wm * TM + fr, ks * 4 + fq);
    b_off[ks] = A_BYTES + lds_off(wn * TN + fr, ks * 4 + fq);
  }
  auto frag = [&](const char* sb, int off, int t) { return *reinterpret_cast<const bf16x8*>(sb + off + t * 16 * ROW_BYTES); };

#define W4_FENCE() __builtin_amdgcn_sched_barrier(0)
#define W4_BARRIER() do { W4_FENCE(); __builtin_amdgcn_s_barrier(); W4_FENCE(); } while (0)

sk_segment: __attribute__((unused));                  // stream-K: every further segment of this workgroup's range re-enters here
  select(0);
#pragma unroll
  for (int i = 0; i < A_LD + B_LD; ++i) dma(0, i);
  if (nk > 1) {
    select(1);
#pragma unroll
    for (int i = 0; i < A_LD + B_LD; ++i) dma(1, i);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND) : "memory");      // tile 0 landed; tile 1 (ND younger loads) may still fly
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  W4_BARRIER();

  // Fragment registers: W fragments are double-buffered (wf[0], wf[1]); A fragments are refreshed IN
  // PLACE - with the i-major MFMA order af[i] dies after groups 2i and 2i+1, so the next k-step's af[i]
  // is loaded right behind it.  Only af[7] dies too late (it would have to be read after the barrier that
  // releases its buffer), so it alone has a second slot (a7[2]).
  bf16x8 wf[2][NT], af[MT - 1], a7[2];
  // Row-split loop (ROWSPLIT): one set of fragments for BOTH k-steps, wr[ks][j] / ar[ks][i]; see tile_r below.
  constexpr int HM = MT / 2;
  bf16x8 wr[2][NT], ar[2][MT];
  if constexpr (!ROWSPLIT) {
#pragma unroll
    for (int j = 0; j < NT; ++j) wf[0][j] = frag(smem, b_off[0], j);
#pragma unroll
    for (int i = 0; i < MT - 1; ++i) af[i] = frag(smem, a_off[0], i);
    a7[0] = frag(smem, a_off[0], MT - 1);
  } else {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int j = 0; j < NT - 1; ++j) wr[ks][j] = frag(smem, b_off[ks], j);      // (fragment NT-1 is fetched at the top of every tile)
#pragma unroll
      for (int i = 0; i < HM; ++i) ar[ks][i] = frag(smem, a_off[ks], i);
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);        // so that no compiler-inserted wait lands inside the loop
  W4_FENCE();

  // one K-tile; MORE: a tile kt+1 exists (fetch its first fragments), LOAD2: tile kt+2 exists (LDS-DMA it)
  auto tile_k = [&](int kt, auto more_c, auto load2_c) __attribute__((always_inline)) {
    constexpr bool MORE = decltype(more_c)::value, LOAD2 = decltype(load2_c)::value;
    const int buf = PERSIST ? p_par : (kt & 1);            // the stream's K-tiles alternate stages across output tiles
    const char* sb = smem + buf * STAGE;
    const char* sn = smem + (buf ^ 1) * STAGE;
    // ---------------- block 1: MFMA(k-step 0) || reads of k-step 1
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (!(ABL & 2) || kt == 0) {
        if (g < NT) wf[1][g] = frag(sb, b_off[1], g);
        // af[i] dies with MFMA i*NT + NT-1, i.e. in group (i*NT + NT-1)/4; it is refetched in the next group
        if (g >= 1 && (g * 4) % NT == 0 && g * 4 / NT - 1 <= MT - 2) af[g * 4 / NT - 1] = frag(sb, a_off[1], g * 4 / NT - 1);
        if (g == LAST_A) a7[1] = frag(sb, a_off[1], MT - 1);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int t = g * 4 + q, i = t / NT, j = t % NT;
        mfma_tied(acc[i][j], wf[0][j], i == MT - 1 ? a7[0] : af[i]);
      }
      W4_FENCE();
    }
    // every wave is done reading buffer kt&1 (lgkmcnt) and its pieces of tile kt+1 have landed (vmcnt)
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if constexpr (!(ABL & 8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (!(ABL & 4)) { W4_BARRIER(); } else { W4_FENCE(); }
    if constexpr (LOAD2) { if constexpr (PERSIST) select_p(kt + 2); else select(kt + 2); }
    // ---------------- block 2: MFMA(k-step 1) || DMA of tile kt+2 || reads of tile kt+1, k-step 0
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if constexpr (LOAD2 && !(ABL & 1)) {
        dma(buf, g);
        if (g + NG < ND) dma(buf, g + NG);
      }
      if constexpr (MORE && !(ABL & 2)) {
        if (g < NT) wf[0][g] = frag(sn, b_off[0], g);
        if (g >= 1 && (g * 4) % NT == 0 && g * 4 / NT - 1 <= MT - 2) af[g * 4 / NT - 1] = frag(sn, a_off[0], g * 4 / NT - 1);
        if (g == LAST_A) a7[0] = frag(sn, a_off[0], MT - 1);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int t = g * 4 + q, i = t / NT, j = t % NT;
        mfma_tied(acc[i][j], wf[1][j], i == MT - 1 ? a7[1] : af[i]);
      }
      W4_FENCE();
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);      // all of the next block's fragments are in (issued >= 2 groups ago)
    // accumulator copies the register allocator may place on the loop's exit edges must not overtake the
    // last (asm, hence opaque) MFMAs: XDL-write -> VALU-read needs up to 18 wait states
    asm volatile("s_nop 15" ::: "memory");
    W4_FENCE();
    if constexpr (PERSIST) p_par ^= 1;
  };
  // Row-split K-tile (the MX-fp8 kernel's schedule, gemm_fp8.hip): the tile's MFMAs run as two blocks of NT groups, group j =
  // W fragment j (both k-steps) against row tiles 0..HM-1 (block 1) or HM..MT-1 (block 2).  W fragments are refreshed in place
  // one group behind their last use in block 2, the row halves alternate, so ONE set of fragments serves and - unlike the
  // k-step split, whose stage only frees up at the middle barrier - the W image is free early in block 1 (barrier S): the
  // LDS-DMA pieces of tile kt+2 are spread over BOTH blocks (W in block 1, A in block 2), half as dense among the MFMAs.
#ifdef VLB_TOOLS
  auto tile_r = [&](int kt, auto more_c, auto load2_c) __attribute__((always_inline)) {
    constexpr bool MORE = decltype(more_c)::value, LOAD2 = decltype(load2_c)::value;
    constexpr int S_AT = 1;
    const char* sb = smem + (kt & 1) * STAGE;
    const char* sn = smem + ((kt & 1) ^ 1) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wr[ks][NT - 1] = frag(sb, b_off[ks], NT - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = HM; i < MT; ++i) ar[ks][i] = frag(sb, a_off[ks], i);
    // ---------------- block 1: rows 0..HM-1 || DMA of W(kt+2)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if constexpr (LOAD2) {
        if (j == S_AT) {                      // S: every wave holds all W fragments of tile kt - this stage's W image is free
          __builtin_amdgcn_s_waitcnt(0xc07f);
          W4_BARRIER();
          select(kt + 2);
        }
        if (j >= S_AT) {
#pragma unroll
          for (int pc = 0; pc < B_LD; ++pc)
            if (pc * (NT - S_AT) / B_LD == j - S_AT) dma(kt & 1, A_LD + pc);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < HM; ++i) mfma_tied(acc[i][j], wr[ks][j], ar[ks][i]);
      W4_FENCE();
    }
    if constexpr (MORE) {
      // M: every wave holds all of tile kt (the A image is free too); tile kt+1 has landed - only W(kt+2) may still fly
      __builtin_amdgcn_s_waitcnt(0xc07f);
      if constexpr (LOAD2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(B_LD) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      W4_BARRIER();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < HM; ++i) ar[ks][i] = frag(sn, a_off[ks], i);
    }
    // ---------------- block 2: rows HM..MT-1 || DMA of A(kt+2) || W fragments of tile kt+1 in place
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if constexpr (LOAD2) {
#pragma unroll
        for (int pc = 0; pc < A_LD; ++pc)
          if (pc * NT / A_LD == j) dma(kt & 1, pc);
      }
      if constexpr (MORE) {
        if (j >= 1) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) wr[ks][j - 1] = frag(sn, b_off[ks], j - 1);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = HM; i < MT; ++i) mfma_tied(acc[i][j], wr[ks][j], ar[ks][i]);
      W4_FENCE();
    }
    asm volatile("s_nop 15" ::: "memory");
    W4_FENCE();
  };
  auto tile = [&](int kt, auto more_c, auto load2_c) __attribute__((always_inline)) {
    if constexpr (ROWSPLIT) tile_r(kt, more_c, load2_c); else tile_k(kt, more_c, load2_c);
  };
#else
  auto& tile = tile_k;                   // product build: the k-step loop, called directly (tile_r is never instantiated)
#endif
  using T_ = std::true_type; using F_ = std::false_type;
  if constexpr (PERSIST) {
    // Persistent stream (tools experiment): the K loop runs on across output tiles - the last two K-tiles of a tile issue
    // the LDS-DMA of the next tile's first two, the last one fetches its first fragments - so between two tiles only the
    // epilogue (convert + store, no wait) and the re-zeroing of the accumulators stand in the MFMA stream: no workgroup
    // launch, no cold prologue, no drain.
    const int fq_ = lane >> 4;
    auto epilogue_p = [&]() __attribute__((always_inline)) {
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; j += 4)
          asm volatile("" : "+a"(acc[i][j]), "+a"(acc[i][j + 1]), "+a"(acc[i][j + 2]), "+a"(acc[i][j + 3]));
      if (p.act == VLB_ACT_SWIGLU_PAIR) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int m = m0 + wm * TM + i * 16 + fr;
          if (m >= p.M) continue;
#pragma unroll
          for (int j = 0; j < NT; j += 4)
            store_swiglu8(p, acc[i][j], acc[i][j + 1], acc[i][j + 2], acc[i][j + 3], m, (n0 + wn * TN) / 2 + (j / 2) * 16 + fq_ * 4, fq_);
        }
      } else {
#define W4_EPILOGUE_P(KIND)                                                                                            \
  _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                                      \
    const int m = m0 + wm * TM + i * 16 + fr;                                                                           \
    if (m >= p.M) continue;                                                                                             \
    _Pragma("unroll") for (int j = 0; j < NT; j += 2)                                                                   \
      w4_store_frag2<KIND>(p, acc[i][j], acc[i][j + 1], m, n0 + wn * TN + j * 16 + fq_ * 4, fq_);                       \
  }
        const int kind = w4_epilogue_kind(p);
        if (kind == EPI_PLAIN) { W4_EPILOGUE_P(EPI_PLAIN) }
        else if (kind == EPI_RESIDUAL) { W4_EPILOGUE_P(EPI_RESIDUAL) }
        else { W4_EPILOGUE_P(EPI_GENERIC + 0) }              // (bias / no activation; the tools launch admits nothing else)
#undef W4_EPILOGUE_P
      }
    };
    for (p_it = 0; p_it < p.persist_iters; ++p_it) {
      const bool last_tile = p_it + 1 == p.persist_iters;
      const int n_tt = last_tile ? nk - 2 : nk;
      for (int kt = 0; kt < n_tt; ++kt) tile(kt, T_{}, T_{});
      if (last_tile) {
        tile(nk - 2, T_{}, F_{});
        tile(nk - 1, F_{}, F_{});
      }
      epilogue_p();
      if (p_it + 1 < p.persist_iters) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; asm volatile("" : "+a"(acc[i][j])); }
        asm volatile("s_nop 7" ::: "memory");                // VALU writes of the accumulators settle before the next MFMA reads them
        m0 = nm0; n0 = nn0;
      }
    }
    return;
  }
  int kt = 0;
  if constexpr (MASKED) {
    // K-tile 0 = the LoRA pair u.A (host guarantees K2 == 64 and nk >= 3); then keep/(1-p) on the accumulators.
    // In a split-K tail only the first K range holds the pair: the others run their first main tile here and the
    // mask degenerates to keep-all x 1.0 (same straight-line code, no branch around the pinned accumulators).
    const bool later_range = (SPLITK && ksplit != 0) || (STREAMK && kt_lo != 0);
    const uint32_t m_thresh = later_range ? 0u : p.drop_thresh;
    const float m_scale = later_range ? 1.f : p.drop_scale;
    tile(0, T_{}, T_{});
    kt = 1;
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; j += 4)
        asm volatile("" : "+a"(acc[i][j]), "+a"(acc[i][j + 1]), "+a"(acc[i][j + 2]), "+a"(acc[i][j + 3]));
    const int frm = lane & 15, fqm = lane >> 4;
    const uint32_t halfN = (uint32_t)(p.N >> 1);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const uint32_t rb = (uint32_t)(m0 + wm * TM + i * 16 + frm) * halfN;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const uint32_t pair0 = (uint32_t)(n0 + wn * TN + j * 16 + fqm * 4) >> 1;
        f32x4 v = acc[i][j];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          uint32_t h = (rb + pair0 + q) ^ p.drop_key;
          h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16;      // lowbias32
          v[2 * q] = (h & 0xffffu) >= m_thresh ? v[2 * q] * m_scale : 0.f;
          v[2 * q + 1] = (h >> 16) >= m_thresh ? v[2 * q + 1] * m_scale : 0.f;
        }
        acc[i][j] = v;
        // straight back into its AGPR tuple, one tile at a time: keeps the VGPR live set (fragments of the next
        // K-tile are already in flight) small and the allocator from parking accumulators in VGPRs
        asm volatile("" : "+a"(acc[i][j]));
        W4_FENCE();
      }
    }
    asm volatile("s_nop 7" ::: "memory");          // VALU writes of the accumulators settle before the next MFMA reads them
  }
  for (; kt + 2 < nk; ++kt) tile(kt, T_{}, T_{});
  if (kt + 1 < nk) { tile(kt, T_{}, F_{}); ++kt; }
  tile(kt, F_{}, F_{});
  // The compiler does not know the asm statements are MFMAs: left alone it reads accumulators
  // (v_accvgpr_read for the epilogue) one cycle after the MFMA that produces them.  Re-define every
  // accumulator tile in (empty) asm statements placed after the wait, so all reads are ordered behind it.
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; j += 4)
      asm volatile("" : "+a"(acc[i][j]), "+a"(acc[i][j + 1]), "+a"(acc[i][j + 2]), "+a"(acc[i][j + 3]));
#undef W4_FENCE
#undef W4_BARRIER

  if constexpr (SPLITK) {
    // work item (= remapped block id) ksplit * tail_tiles + tile owns one [4 waves][MT][NT][64 lanes][4] slab
    float* wt = p.ws + ((int64_t)xcd_remap_pid(p.order) * 4 + wave) * (MT * NT * 256);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        *reinterpret_cast<f32x4*>(wt + ((i * NT + j) * 64 + lane) * 4) = acc[i][j];
    return;
  }
#define W4_DONE do { if constexpr (STREAMK) goto sk_next; else return; } while (0)
  if constexpr (STREAMK) {
    if (kt_lo != 0) {
      // contributor: the rest of a tile whose first K-tiles belong to workgroup sk_r - 1.  Slab sk_r, then the flag.
      float* wt = p.ws + ((int64_t)sk_r * 4 + wave) * (MT * NT * 256);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          *reinterpret_cast<f32x4*>(wt + ((i * NT + j) * 64 + lane) * 4) = acc[i][j];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's slab stores have reached L2
      __builtin_amdgcn_s_barrier();                             // ... and so have the other three waves'
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // L2 write-back (buffer_wbl2 sc1): visible to the owner's XCD
        __hip_atomic_store(p.sk_flags + sk_r, p.sk_want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      goto sk_next;
    }
    if (nk < nk_all) {
      // owner of a tile whose remaining K-tiles are the first segment of workgroup sk_r + 1: wait for its slab (bounded),
      // add it to the accumulators, then the normal epilogue
      if (tid == 0) {
        int spins = 0;
        while (__hip_atomic_load(p.sk_flags + sk_r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.sk_want) {
          __builtin_amdgcn_s_sleep(16);
          if (++spins > (1 << 20)) {                            // ~2 s: never in a healthy launch; results are then wrong and the host is told
            __hip_atomic_store(p.sk_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
          }
        }
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");        // L2 / L1 invalidate (buffer_inv sc1): the slab is read from where the release put it
      const float* ps = p.ws + ((int64_t)(sk_r + 1) * 4 + wave) * (MT * NT * 256);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(ps + ((i * NT + j) * 64 + lane) * 4);
          f32x4 v = acc[i][j];
          v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
          acc[i][j] = v;
          asm volatile("" : "+a"(acc[i][j]));
        }
    }
  }
  // The four-wave kernels store 16 bytes per lane (store_pair16): the host only routes launches here whose C / aux rows are
  // 16-byte aligned (GemmArgs::wide).  One loop nest per epilogue kind - every loop over acc[][] must unroll completely
  // (a dynamically indexed accumulator array is moved to scratch memory, and with it the whole K loop).
  if (p.act == VLB_ACT_SWIGLU_PAIR) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + wm * TM + i * 16 + fr;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NT; j += 4)
        store_swiglu8(p, acc[i][j], acc[i][j + 1], acc[i][j + 2], acc[i][j + 3], m, (n0 + wn * TN) / 2 + (j / 2) * 16 + fq * 4, fq);
    }
    W4_DONE;
  }
#define W4_EPILOGUE(KIND)                                                                                              \
  _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                                      \
    const int m = m0 + wm * TM + i * 16 + fr;                                                                           \
    if (m >= p.M) continue;                                                                                             \
    _Pragma("unroll") for (int j = 0; j < NT; j += 2)                                                                   \
      w4_store_frag2<KIND>(p, acc[i][j], acc[i][j + 1], m, n0 + wn * TN + j * 16 + fq * 4, fq);                         \
  }
  {
    const int kind = w4_epilogue_kind(p);
    if (kind == EPI_PLAIN) { W4_EPILOGUE(EPI_PLAIN) W4_DONE; }
    if (kind == EPI_RESIDUAL) { W4_EPILOGUE(EPI_RESIDUAL) W4_DONE; }
    if (kind == EPI_SWIGLU_BWD) { W4_EPILOGUE(EPI_SWIGLU_BWD) W4_DONE; }
#define W4_EPILOGUE_ACT(ACT) W4_EPILOGUE(EPI_GENERIC + ACT)
    VLB_DISPATCH_ACT(p.act, W4_EPILOGUE_ACT);
#undef W4_EPILOGUE_ACT
#undef W4_EPILOGUE
  }
sk_next: __attribute__((unused));
  if constexpr (STREAMK) {
    sk_it += nk;
    if (sk_it < sk_end) {
      // next segment of this workgroup's range: a new tile from its first K-tile (whole, or the head this workgroup owns)
      tile_coords(p, p.tile0 + sk_it / nk_all, 0, BM, BN, m0, n0);
      rowA = m0 + r0; rowW = n0 + r0;
      kt_lo = sk_it % nk_all; nk = min(nk_all - kt_lo, sk_end - sk_it);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; asm volatile("" : "+a"(acc[i][j])); }
      asm volatile("s_nop 7" ::: "memory");                // VALU writes of the accumulators settle before the next MFMA reads them
      goto sk_segment;
    }
  }
#undef W4_DONE
}

#ifdef VLB_TOOLS
// ------------------------------------------------------------------------------------------------
// W-direct variant of the four-wave 256x256x64 tile (gemm_wd_kernel).  TOOLS BUILD ONLY: built in round 4 because the
// ablations of gemm_w4_kernel pointed at the LDS-DMA issue, measured 4-9 % SLOWER than gemm_w4_kernel on every shape it serves
// (interleaved same-process A/B, outputs bit-identical; DESIGN.md 5.1 has the table and the instruction-level probe that
// explains it: a fragment-shaped global load - 16 rows x 64 B - costs the vector memory path ~4x a line-shaped one per byte).
// Kept as the record of that experiment and as the A/B partner of tools/ab_gemm_rowsplit.py (VLB_AB=wd).
//
// What the ablations of gemm_w4_kernel say: with one wave per SIMD the LDS-DMA ISSUE is the largest cost
// of the K loop (16 pieces per wave per K-tile, 60-185 issue cycles each, during which the in-order wave issues no MFMA).
// Here only A goes through LDS: the four waves sit 1 (M) x 4 (N), wave w owns output columns [64w, 64w+64) of ALL 256
// rows (16 x 4 accumulator tiles = the same 256 AGPRs), so
//   * a wave's W fragments are its own (no other wave needs them): they come straight from global / L2 into REGISTERS,
//     one K-tile ahead - 8 plain global_load_dwordx4 per wave per K-tile (16 rows x 64 B each: lane (fr, fq) fetches the
//     16 bytes of W row fr that it feeds to the MFMA as k-group fq), no LDS write, no LDS read, no barrier dependence;
//   * A (the operand all four waves share) is staged by LDS-DMA as before: 32 pieces of 1 KiB per K-tile = 8 per wave
//     (half the pieces of the 2 x 2 layout), 32 KB per stage, two stages;
//   * every wave reads all 16 A fragments of a k-step from LDS: 32 ds_read_b128 per wave per K-tile, as many as before,
//     against an LDS image that now takes half the DMA write traffic.
// Per K-tile and wave: 128 MFMA, 32 ds_read_b128, 8 LDS-DMA pieces, 8 global loads, 1 barrier, two COUNTED vmcnt waits
// (never 0 in the loop: W(t+1) is waited for with A(t+2)'s pieces still in flight and vice versa).
//
// Registers by hand.  The W double buffer lives in v[192:255]: buffer b, k-step ks, fragment j = v[192 + 16(2b+ks)... see
// WD_WREG.  The loads and the MFMAs that read them name those registers literally (the compiler-allocated prototype of
// round 2 died in register allocation: asm outputs that are written asynchronously cannot be expressed as operands); every
// asm statement that writes them lists all 64 as clobbers, so the compiler keeps nothing of its own there across any of
// them, and tools/audit_gemm_isa.py proves on the generated ISA that no compiler-issued instruction touches v192+ between
// the first W load and the end of the K loop.  Accumulators are tied AGPR tuples as in gemm_w4_kernel.
// ------------------------------------------------------------------------------------------------
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
#define WD_CLOBBER "v192","v193","v194","v195","v196","v197","v198","v199","v200","v201","v202","v203","v204","v205","v206","v207","v208","v209","v210","v211","v212","v213","v214","v215","v216","v217","v218","v219","v220","v221","v222","v223","v224","v225","v226","v227","v228","v229","v230","v231","v232","v233","v234","v235","v236","v237","v238","v239","v240","v241","v242","v243","v244","v245","v246","v247","v248","v249","v250","v251","v252","v253","v254","v255"
// W register quad of buffer b (tile parity), k-step ks, fragment j
constexpr int WD_WREG(int b, int ks, int j) { return 192 + ((b * 2 + ks) * 4 + j) * 4; }
template <int R, int OFF>
__device__ __forceinline__ void wd_load_w(uint32_t voff, const char* sbase) {
  asm volatile("global_load_dwordx4 v[%c2:%c3], %0, %1 offset:%c4" ::"v"(voff), "s"(sbase), "i"(R), "i"(R + 3), "i"(OFF) : "memory", WD_CLOBBER);
}
template <int R>
__device__ __forceinline__ void wd_mfma(f32x4& c, const bf16x8& a) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, v[%c2:%c3], %1, %0" : "+a"(c) : "v"(__builtin_bit_cast(i32x4_t, a)), "i"(R), "i"(R + 3));
}

__global__ __launch_bounds__(256, 1) void gemm_wd_kernel(GemmArgs p) {
  constexpr int MT = 16, NT = 4, BM = 256, BN = 256, TN = 16 * NT;
  constexpr int STAGE = BM * ROW_BYTES;               // one A tile: 32 KB
  constexpr int A_LD = BM / 32;                       // LDS-DMA pieces per wave per K-tile
  constexpr int W_LD = 2 * NT;                        // W loads per wave per K-tile
  constexpr int LAST_A = MT / 2;                      // group in which the double-slotted last A fragment is fetched
  extern __shared__ __attribute__((aligned(16))) char smem[];
  asm volatile("" ::: WD_CLOBBER);                    // v[192:255] are part of this kernel's register budget from here on

  int m0, n0;
  map_tile(p, BM, BN, m0, n0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;

  // A staging as in gemm_w4_kernel: piece i of this wave fills LDS rows [i*32 + wave*8, +8), swizzle on the source address
  const int srow = (lane >> 3), sslot = lane & 7;
  const int r0 = wave * 8 + srow;
  const int colb = (sslot ^ ((r0 >> 1) & 7)) * 8;
  const int rowA = m0 + r0;
  const int rowW = n0 + wave * TN + fr;               // W row of this lane in fragment 0 (fragment j: + 16 j)
  const int nk1 = p.K / BK, nk = nk1 + p.K2 / BK;

  // two cursors: A's LDS-DMA runs two K-tiles ahead of the MFMAs, W's register loads one
  uint32_t offA[A_LD], offW[NT];
  const char* curA; const char* curW;
  auto set_a = [&](const bf16* A_, int lda_) {
#pragma unroll
    for (int i = 0; i < A_LD; ++i) offA[i] = ((uint32_t)min(rowA + 32 * i, p.M - 1) * (uint32_t)lda_ + (uint32_t)colb) * 2u;
    curA = reinterpret_cast<const char*>(A_);
  };
  auto set_w = [&](const bf16* W_, int ldw_) {
#pragma unroll
    for (int j = 0; j < NT; ++j) offW[j] = ((uint32_t)(rowW + 16 * j) * (uint32_t)ldw_ + (uint32_t)(fq * 8)) * 2u;
    curW = reinterpret_cast<const char*>(W_);
  };
  auto select_a = [&](int kt) { if (kt == 0) set_a(p.A, p.lda); else if (kt == nk1) set_a(p.A2, p.lda2); else curA += ROW_BYTES; };
  auto select_w = [&](int kt) { if (kt == 0) set_w(p.W, p.ldw); else if (kt == nk1) set_w(p.W2, p.ldw2); else curW += ROW_BYTES; };
  auto dma = [&](int buf, int i) { glds16(curA + offA[i], smem + buf * STAGE + wave * 8 * ROW_BYTES + i * 32 * ROW_BYTES); };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int a_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) a_off[ks] = lds_off(fr, ks * 4 + fq);
  auto frag = [&](const char* sb, int off, int t) { return *reinterpret_cast<const bf16x8*>(sb + off + t * 16 * ROW_BYTES); };

#define WD_FENCE() __builtin_amdgcn_sched_barrier(0)
#define WD_BARRIER() do { WD_FENCE(); __builtin_amdgcn_s_barrier(); WD_FENCE(); } while (0)

  // prologue: A(0) -> stage 0, W(0) -> register buffer 0, A(1) -> stage 1
  select_a(0);
#pragma unroll
  for (int i = 0; i < A_LD; ++i) dma(0, i);
  select_w(0);
  static_for<0, W_LD>([&](auto qc) {
    constexpr int q = decltype(qc)::value, j = q >> 1, ks = q & 1;
    wd_load_w<WD_WREG(0, ks, j), ks * 64>(offW[j], curW);
  });
  if (nk > 1) select_a(1);                            // (a single-tile K range re-fetches tile 0: same counted waits everywhere)
#pragma unroll
  for (int i = 0; i < A_LD; ++i) dma(1, i);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LD) : "memory");        // A(0) and W(0) are in; A(1) may still fly
  WD_BARRIER();

  // A fragments are refreshed IN PLACE: af[i] is consumed by the 4 MFMAs of group i and refetched (next k-step) at the top
  // of group i+1; the last one would be refetched right in front of the barrier that needs it complete, so it has two slots.
  bf16x8 af[MT - 1], al[2];
#pragma unroll
  for (int i = 0; i < MT - 1; ++i) af[i] = frag(smem, a_off[0], i);
  al[0] = frag(smem, a_off[0], MT - 1);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  WD_FENCE();

  // One K-tile kt of parity PAR (= its LDS stage and its W register buffer).  ONE body serves every tile: past the end of
  // the K range the cursors simply stop advancing, so the last tiles re-fetch the final K-tile's operands into buffers that
  // nobody reads again (redundant loads of valid addresses instead of tail variants of a 128-MFMA body: with several
  // variants inlined the register allocator no longer keeps the 64 accumulator tuples in place).
  auto tile = [&](int kt, auto par_c) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;
    const char* sb = smem + PAR * STAGE;
    const char* sn = smem + (PAR ^ 1) * STAGE;
    if (kt + 1 < nk) select_w(kt + 1);
    // ---------------- block 1: MFMA(k-step 0) || A reads of k-step 1 || W(kt+1) -> the other register buffer
    static_for<0, MT>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i >= 1) af[i - 1] = frag(sb, a_off[1], i - 1);
      if constexpr (i == LAST_A) al[1] = frag(sb, a_off[1], MT - 1);
      if constexpr (i % 2 == 0) {
        constexpr int q = i / 2, j = q >> 1, ks = q & 1;
        wd_load_w<WD_WREG(PAR ^ 1, ks, j), ks * 64>(offW[j], curW);
      }
      static_for<0, NT>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (i == MT - 1) wd_mfma<WD_WREG(PAR, 0, j)>(acc[i][j], al[0]);
        else wd_mfma<WD_WREG(PAR, 0, j)>(acc[i][j], af[i]);
      });
      WD_FENCE();
    });
    // this wave is done reading stage PAR (lgkmcnt); A(kt+1) has landed: only the W(kt+1) loads, issued after it, may fly
    __builtin_amdgcn_s_waitcnt(0xc07f);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_LD) : "memory");
    WD_BARRIER();
    if (kt + 2 < nk) select_a(kt + 2);
    // ---------------- block 2: MFMA(k-step 1) || LDS-DMA of A(kt+2) into stage PAR || A reads of tile kt+1, k-step 0
    static_for<0, MT>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i >= 1) af[i - 1] = frag(sn, a_off[0], i - 1);
      if constexpr (i == LAST_A) al[0] = frag(sn, a_off[0], MT - 1);
      if constexpr (i % 2 == 0) dma(PAR, i / 2);
      static_for<0, NT>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (i == MT - 1) wd_mfma<WD_WREG(PAR, 1, j)>(acc[i][j], al[1]);
        else wd_mfma<WD_WREG(PAR, 1, j)>(acc[i][j], af[i]);
      });
      WD_FENCE();
    });
    // next tile's first fragments are in; W(kt+1) has landed: only A(kt+2)'s pieces, issued after it, may fly
    __builtin_amdgcn_s_waitcnt(0xc07f);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LD) : "memory");
    asm volatile("s_nop 15" ::: "memory");      // XDL write -> (compiler-placed) VALU read of an accumulator on a loop exit edge
    WD_FENCE();
  };
  using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) { tile(kt, P0{}); tile(kt + 1, P1{}); }
  if (kt < nk) tile(kt, P0{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the redundant tail fetches (register and LDS destinations) are retired
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
  for (int i = 0; i < MT; ++i) asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]));
#undef WD_FENCE
#undef WD_BARRIER

  // epilogues of gemm_w4_kernel with this kernel's wave block (rows m0.., columns n0 + 64 wave ..)
  if (p.act == VLB_ACT_SWIGLU_PAIR) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + i * 16 + fr;
      if (m >= p.M) continue;
      store_swiglu8(p, acc[i][0], acc[i][1], acc[i][2], acc[i][3], m, (n0 + wave * TN) / 2 + fq * 4, fq);
    }
    return;
  }
#define WD_EPILOGUE(KIND)                                                                                              \
  _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                                      \
    const int m = m0 + i * 16 + fr;                                                                                     \
    if (m >= p.M) continue;                                                                                             \
    _Pragma("unroll") for (int j = 0; j < NT; j += 2)                                                                   \
      w4_store_frag2<KIND>(p, acc[i][j], acc[i][j + 1], m, n0 + wave * TN + j * 16 + fq * 4, fq);                       \
  }
  const int kind = w4_epilogue_kind(p);
  if (kind == EPI_PLAIN) { WD_EPILOGUE(EPI_PLAIN) return; }
  if (kind == EPI_RESIDUAL) { WD_EPILOGUE(EPI_RESIDUAL) return; }
  if (kind == EPI_SWIGLU_BWD) { WD_EPILOGUE(EPI_SWIGLU_BWD) return; }
#define WD_EPILOGUE_ACT(ACT) WD_EPILOGUE(EPI_GENERIC + ACT)
  VLB_DISPATCH_ACT(p.act, WD_EPILOGUE_ACT);
#undef WD_EPILOGUE_ACT
#undef WD_EPILOGUE
}

#endif  // VLB_TOOLS (W-direct experiment)

// Second half of a split-K tail: block (tile, i) sums row-tile i of every wave's accumulators over the splits in
// a fixed order (deterministic) and applies the four-wave kernel's epilogue with the same thread <-> element map.
template <int MT, int NT>
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmArgs p) {
  constexpr int TM = 16 * MT, TN = 16 * NT, BM = 2 * TM, BN = 2 * TN;
  const int tl = blockIdx.x / MT, i = blockIdx.x % MT;
  int m0, n0;
  tile_coords(p, p.tile0 + tl, 0, BM, BN, m0, n0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
  f32x4 v[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < p.k_splits; ++s) {
    const float* wt = p.ws + ((int64_t)(s * p.tail_tiles + tl) * 4 + wave) * (MT * NT * 256);
#pragma unroll
    for (int j = 0; j < NT; ++j) v[j] += *reinterpret_cast<const f32x4*>(wt + ((i * NT + j) * 64 + lane) * 4);
  }
  const int m = m0 + wm * TM + i * 16 + fr;
  if (m >= p.M) return;
  if (p.act == VLB_ACT_SWIGLU_PAIR) {
#pragma unroll
    for (int j = 0; j < NT; j += 4) store_swiglu8(p, v[j], v[j + 1], v[j + 2], v[j + 3], m, (n0 + wn * TN) / 2 + (j / 2) * 16 + fq * 4, fq);
    return;
  }
  const int kind = w4_epilogue_kind(p);
#define W4_REDUCE_EPILOGUE(KIND) \
  _Pragma("unroll") for (int j = 0; j < NT; j += 2) w4_store_frag2<KIND>(p, v[j], v[j + 1], m, n0 + wn * TN + j * 16 + fq * 4, fq);
  if (kind == EPI_PLAIN) { W4_REDUCE_EPILOGUE(EPI_PLAIN) return; }
  if (kind == EPI_RESIDUAL) { W4_REDUCE_EPILOGUE(EPI_RESIDUAL) return; }
  if (kind == EPI_SWIGLU_BWD) { W4_REDUCE_EPILOGUE(EPI_SWIGLU_BWD) return; }
#define W4_REDUCE_EPILOGUE_ACT(ACT) W4_REDUCE_EPILOGUE(EPI_GENERIC + ACT)
  VLB_DISPATCH_ACT(p.act, W4_REDUCE_EPILOGUE_ACT);
#undef W4_REDUCE_EPILOGUE_ACT
#undef W4_REDUCE_EPILOGUE
}

#ifdef VLB_TOOLS
int g_w4_rowsplit = 0;              // tools: 1 = every four-wave launch uses the row-split K loop (ABL bit 6)
int g_w4_persist = 0;               // tools: 1 = whole-tile 256-column launches run as a persistent stream (ABL bit 7)
#endif
#ifdef VLB_TOOLS
int g_wd = 0;                       // tools: 1 = whole 256x256-tile launches run on gemm_wd_kernel (A/B of the W-direct experiment)
// whole 256x256 tiles, both operand pairs, every epilogue kind; no masked pair, no split-K (those launches keep gemm_w4_kernel)
inline int launch_wd(GemmArgs& a, hipStream_t s) {
  constexpr int LDS = 2 * 256 * ROW_BYTES;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_wd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  if (attr != hipSuccess) {
    vlb_set_error("gemm: cannot reserve %d bytes of LDS: %s", LDS, hipGetErrorString(attr));
    return VLB_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(gemm_wd_kernel, dim3(a.grid), dim3(256), LDS, s, a);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
#endif

template <int NT, int ABL, int MT = 8, bool MASKED = false, bool SPLITK = false>
int launch_w4(GemmArgs& a, hipStream_t s) {
  constexpr bool STREAMK = (ABL & 256) != 0;
#ifdef VLB_TOOLS
  if constexpr (NT == 8 && ABL == 0 && MT == 8 && !MASKED && !SPLITK && !STREAMK) {
    if (g_wd && a.split_n == 1 && a.k_splits <= 1) return launch_wd(a, s);
  }
#endif
  constexpr int LDS = 2 * (32 * MT + 32 * NT) * ROW_BYTES;
  // once per process and kernel; a function-local static's initialisation is thread-safe (C++11)
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w4_kernel<NT, ABL, MT, MASKED, SPLITK>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  if (attr != hipSuccess) {
    vlb_set_error("gemm: cannot reserve %d bytes of LDS: %s", LDS, hipGetErrorString(attr));
    return VLB_ERR_LAUNCH;
  }
#ifdef VLB_TOOLS
  if constexpr (ABL == 0 && NT == 8 && !MASKED && !SPLITK && !STREAMK) {
    // tools A/B: persistent stream over the full rounds of the launch (the ragged rest as a normal launch)
    const int nk_all = a.K / 64 + a.K2 / 64;
    const bool act_ok = a.act == VLB_ACT_NONE || a.act == VLB_ACT_SWIGLU_PAIR;
    if (g_w4_persist && (a.order & 2) && !(a.order & 4) && a.split_n == 1 && a.k_splits <= 1 && a.grid >= 512 && nk_all >= 4 && act_ok &&
        (a.K2 == 0 || a.K2 >= 64)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w4_kernel<NT, 128, MT, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
        return VLB_ERR_LAUNCH;
      GemmArgs q = a;
      q.persist_iters = a.grid / 256;
      hipLaunchKernelGGL((gemm_w4_kernel<NT, 128, MT, false, false>), dim3(256), dim3(256), LDS, s, q);
      VLB_LAUNCH_CHECK();
      const int rest = a.grid - q.persist_iters * 256;
      if (rest == 0) return VLB_OK;
      GemmArgs r = a;
      r.tile0 = a.tile0 + q.persist_iters * 256; r.grid = rest;
      hipLaunchKernelGGL((gemm_w4_kernel<NT, ABL, MT, MASKED, SPLITK>), dim3(r.grid), dim3(256), LDS, s, r);
      VLB_LAUNCH_CHECK();
      return VLB_OK;
    }
  }
  if constexpr (ABL == 0 && NT == 8 && !MASKED && !SPLITK && !STREAMK) {
    if (g_w4_rowsplit) {                     // tools A/B: the row-split K loop
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w4_kernel<NT, 64, MT, MASKED, SPLITK>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
        return VLB_ERR_LAUNCH;
      hipLaunchKernelGGL((gemm_w4_kernel<NT, 64, MT, MASKED, SPLITK>), dim3(a.grid), dim3(256), LDS, s, a);
      VLB_LAUNCH_CHECK();
      return VLB_OK;
    }
  }
#endif
  hipLaunchKernelGGL((gemm_w4_kernel<NT, ABL, MT, MASKED, SPLITK>), dim3(a.grid), dim3(256), LDS, s, a);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

#ifdef VLB_TOOLS
// ---- stream-K launch (see gemm_w4_kernel).  TOOLS BUILD ONLY: built and measured in round 4, 1.4-2x SLOWER than
// the round + tail plans on every ragged shape of the LoRA step - equal-length iteration ranges put the 256 CUs at different
// K offsets of their tiles, so the CUs that share an A or W panel no longer read the same K-slice at the same time, the per-XCD
// L2 stops serving the re-reads, and the launch runs at the fabric's bandwidth instead of the matrix pipe's rate.
// Workspace = 256 slabs of one 256x256 fp32 tile + 256 8-byte flags behind them.
// A flag is "raised" when it holds (magic << 32 | call number): nothing has to be zeroed, and a stale or uninitialised flag
// cannot match.  The timeout word lives in pinned host memory (one per process): a launch that gave up waiting sets it, and
// every later GEMM call fails loudly instead of returning wrong numbers.
constexpr int64_t SK_SLAB_BYTES = 256ll * 256 * 256 * 4;
constexpr int64_t SK_FLAG_BYTES = 256 * 8;
inline int* sk_err_word() {
  static int* w = [] {
    int* q = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&q), 64, hipHostMallocMapped) != hipSuccess || !q) return (int*)nullptr;
    *q = 0;
    return q;
  }();
  return w;
}
inline unsigned long long sk_next_want() {
  static std::atomic<unsigned int> calls{0};
  return (0x56C4B57Bull << 32) | (unsigned long long)(++calls);
}
template <bool MASKED>
int launch_w4_streamk(GemmArgs a, int tiles_m, int tiles_n, int nk_all, void* ws, hipStream_t s) {
  a.tiles_m = tiles_m; a.tiles_n = tiles_n; a.tile0 = 0; a.split_n = 1; a.k_splits = 0; a.tail_tiles = 0; a.grid = 256;
  a.ws = (float*)ws;
  a.sk_total = tiles_m * tiles_n * nk_all;
  a.sk_flags = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ws) + SK_SLAB_BYTES);
  a.sk_want = sk_next_want();
  a.sk_err = sk_err_word();
  return launch_w4<8, 256, 8, MASKED, false>(a, s);
}
#endif

// partial last wave as a split-K pair of launches: a = the launch of the tail's parent tiles (tile0, tail_tiles,
// k_splits, ws set by the caller)
template <int MT, bool MASKED = false>
int launch_w4_splitk(GemmArgs& a, hipStream_t s) {
  a.split_n = 1;
  a.grid = a.k_splits * a.tail_tiles;
  int rc = launch_w4<8, 0, MT, MASKED, true>(a, s);
  if (rc != VLB_OK) return rc;
  hipLaunchKernelGGL((gemm_splitk_reduce_kernel<MT, 8>), dim3(a.tail_tiles * MT), dim3(256), 0, s, a);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

// Kernel selection.  The product library (libvlb.so) is built WITHOUT VLB_TOOLS: the selection below is a
// compile-time constant, no ablation / superseded kernel is instantiated and no switch is exported.
// `make tools` builds libvlb_tools.so with -DVLB_TOOLS for tools/bench_gemm_variants.py & co (A/B and timing-only
// ablations whose results are wrong by construction); nothing under phantom_vlb_amd/, bench.py or tests/ loads it.
#ifdef VLB_TOOLS
#define VLB_TUNABLE
#else
#define VLB_TUNABLE constexpr
#endif
VLB_TUNABLE int g_variant = 3;      // 0: lock-step double buffer, 1: ping-pong wave groups, 2: four-wave 128x128 blocks,
                                    // 3 (default): four-wave kernel for long K (>= 4096), ping-pong otherwise, 192-row tiles
                                    // when the cost model prefers them; 5 (A/B): like 3 but always 192-row tiles for long K
VLB_TUNABLE int g_force_tile = 0;   // 0: heuristic, 1: 256x256, 2: 256x128 (tuning only)
VLB_TUNABLE int g_tail_split = 1;   // split a mostly idle last wave of tiles into 256x128 tiles
VLB_TUNABLE int g_tile_order = 3;   // GemmArgs::order: bit 0 column bands of 8, bit 1 XCD chunks dealt per round, bit 2 16x16 rounds (A/B: variant bits 10-12 XOR 3)
VLB_TUNABLE int g_order_auto = 1;   // pick_order's shape rule (A/B: variant bit 13 disables it)
VLB_TUNABLE int g_stagger = 0;      // tools: start delay (10-ns ticks) of half the first-round workgroups (timing experiment)
VLB_TUNABLE int g_tail_splitk = 1;  // ... or, when the caller passes a workspace, along K (A/B: variant bit 9 disables)
#ifdef VLB_TOOLS
int g_streamk = 0;                  // tools: 1 = ragged multi-round outputs as ONE stream-K launch (vlb_gemm_set_streamk; measured slower, see launch_w4_streamk)
// true when the shape should run as ONE stream-K launch of 256-row tiles: more than one round, not a whole number of rounds,
// few enough rounds that the ragged end matters (>= 8 full rounds: the round + tail plans lose < 2 %), every segment long enough
inline bool sk_wanted(int tiles, int nk_all, int act, bool ws_ok) {
  return g_streamk && ws_ok && sk_err_word() && act != VLB_ACT_SWIGLU_PAIR && tiles > 256 && tiles % 256 != 0 && tiles < 8 * 256 &&
         nk_all >= 16;
}
#endif

// Tile order for a shape.  The order decides which operand is swept once and which is re-read once per band, i.e. which
// one has to come back out of the 256 MB Infinity Cache: order 3 (column bands: W once, A re-read per band) unless A is the
// larger operand AND too large to stay cached (> 128 MB), then order 2 (row bands dealt across the XCDs: A once, W re-read).
// Measured (tools/bench_gemm_variants.py): gate/up at M = 5861 / 9447 / 15630 prefers 3 (+2 / +7 / +4 % over order 0), at
// M = 31260 (A = 256 MB) order 2 (+5 % over 3); `down` (K = 14336) at M >= 9447 (A >= 271 MB) order 2 (+2-4 %).
inline int pick_order(int M, int N, int K) {
  const double a_bytes = 2.0 * M * K, w_bytes = 2.0 * N * K;
  const int o = (a_bytes > w_bytes && a_bytes > 128e6) ? 2 : 3;
  return (g_tile_order == 3 && g_order_auto) ? o : g_tile_order;        // tools build: an explicit A/B order wins (variant bit 13: no shape rule)
}

// How the partial last wave of `tiles` tiles (nk K-tiles each) is run.  Costs are in units of one full wave of
// tiles: whole tiles 1.0; re-cut 256x128 halves 0.62 (measured 0.6-0.8); split-K 1/s plus ~16 K-tiles' worth
// (25-40 us measured) of fixed work (fp32 slab store, second launch, reduce).
struct TailPlan { int mode; int splits; double cost; };       // mode 0: none / whole tiles, 1: halves, 2: split-K
inline TailPlan plan_tail(int tiles, int nk, bool ws_ok, bool allow_halves = true) {
  const int cus = 256, r = tiles % cus;
  TailPlan t{0, 1, r == 0 ? 0.0 : 1.0};
  if (r == 0 || tiles <= cus) return t;
  if (allow_halves && g_tail_split && r <= cus * 5 / 8) t = TailPlan{1, 1, 0.62};
  if (ws_ok && g_tail_splitk) {
    int sp = cus / r; if (sp > 8) sp = 8;
    while (sp > 1 && nk / sp < 8) --sp;
    const double c = 1.0 / sp + 16.0 / nk;
    if (sp >= 2 && c < t.cost) t = TailPlan{2, sp, c};
  }
  return t;
}
// 192- or 256-row tiles for an [M, N] output (N % 256 == 0) on the four-wave kernel: cost = waves x tile rows, the
// smaller tile charged 2 % more; returns true for 192 rows.  Also hands back both tail plans.
inline bool plan_rows(int M, int N, int Ktot, bool ws_ok, TailPlan& p256, TailPlan& p192, bool splitk256 = true) {
  const int cus = 256, tn = N / 256, nk = Ktot / BK;
  const int tiles = ((M + 255) / 256) * tn, tiles192 = ((M + 191) / 192) * tn;
  p256 = plan_tail(tiles, nk, ws_ok && splitk256); p192 = plan_tail(tiles192, nk, ws_ok);
  const double c256 = (tiles / cus + p256.cost) * 256.0, c192 = (tiles192 / cus + p192.cost) * 192.0 * 1.02;
  return tiles > cus && c192 < 0.97 * c256;
}

template <int BM, int BN, int WM, int WN, int S0, int S1, int S2, int S3, int ABL = 0>
int launch_pp(GemmArgs& a, hipStream_t s, int lds) {
  // once per process and kernel; a function-local static's initialisation is thread-safe (C++11)
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_kernel<BM, BN, WM, WN, S0, S1, S2, S3, ABL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (attr != hipSuccess) {
    vlb_set_error("gemm: cannot reserve %d bytes of LDS: %s", lds, hipGetErrorString(attr));
    return VLB_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((gemm_pp_kernel<BM, BN, WM, WN, S0, S1, S2, S3, ABL>), dim3(a.grid), dim3(512), lds, s, a);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

template <int BM, int BN, int WM, int WN>
int launch_tile(GemmArgs& a, hipStream_t s) {
  constexpr int LDS = 2 * (BM + BN) * ROW_BYTES;
#ifdef VLB_TOOLS      // the lock-step kernel (variant 0) only exists in the tools build
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tile_kernel<BM, BN, WM, WN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  if (attr != hipSuccess) {
    vlb_set_error("gemm: cannot reserve %d bytes of LDS: %s", LDS, hipGetErrorString(attr));
    return VLB_ERR_LAUNCH;
  }
#endif
  if (a.grid == 0) {          // plain launch: the whole GEMM with BM x BN tiles
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = a.N / BN;
    a.tile0 = 0; a.split_n = 1;
    a.grid = a.tiles_m * a.tiles_n;
  }
  // the 4-wave kernel addresses operands with 32-bit byte offsets
  // ... and stores 16 bytes per lane (a.wide: C / aux rows 16-byte aligned)
  const bool fits32 = a.wide && (int64_t)a.M * a.lda < (1ll << 31) && (int64_t)a.N * a.ldw < (1ll << 31) &&
                      (int64_t)a.M * a.lda2 < (1ll << 31) && (int64_t)a.N * a.ldw2 < (1ll << 31);
#ifndef VLB_TOOLS
  // the four-wave kernel wins once the K loop is long enough to amortise its serial prologue and epilogue (one
  // workgroup per CU, nothing to overlap them with): measured crossover K ~ 3072-4096
  if constexpr (BM == 256 && BN == 256) {
    if (fits32 && a.K + a.K2 >= 4096) return launch_w4<8, 0>(a, s);
  } else if constexpr (BM == 256 && BN == 128) {
    if (fits32 && a.K + a.K2 >= 4096) return launch_w4<4, 0>(a, s);
  }
  return launch_pp<BM, BN, WM, WN, (BM + BN) / 64, 0, 0, 0>(a, s, LDS);
#else
  if constexpr (BM == 256 && BN == 256) {
    if (!fits32 && (g_variant == 2 || (g_variant >= 0x20 && g_variant < 0x30))) return launch_pp<BM, BN, WM, WN, (BM + BN) / 64, 0, 0, 0>(a, s, LDS);
    if (g_variant == 2) return launch_w4<8, 0>(a, s);
    if (g_variant == 3 || g_variant == 4 || g_variant == 5 || g_variant == 6) {      // 4 (A/B only): four-wave main launch, 8-wave kernel for the re-cut tail
      if (fits32 && a.K + a.K2 >= 4096) return launch_w4<8, 0>(a, s);
      return launch_pp<BM, BN, WM, WN, (BM + BN) / 64, 0, 0, 0>(a, s, LDS);
    }
    if (g_variant == 0x21) return launch_w4<8, 1>(a, s);     // timing-only ablations (wrong results)
    if (g_variant == 0x22) return launch_w4<8, 2>(a, s);
    if (g_variant == 0x24) return launch_w4<8, 4>(a, s);
    if (g_variant == 0x27) return launch_w4<8, 7>(a, s);
    if (g_variant == 0x28) return launch_w4<8, 8>(a, s);
    if (g_variant == 0x41) return launch_w4<8, 16>(a, s);    // W always from panel 0 (L2-resident)
    if (g_variant == 0x42) return launch_w4<8, 32>(a, s);    // A always from panel 0
    if (g_variant == 0x43) return launch_w4<8, 48>(a, s);    // both
  } else {
    if constexpr (BM == 256 && BN == 128) {
      if ((g_variant == 2 || g_variant == 3 || g_variant == 5 || g_variant == 6) && fits32 && a.K + a.K2 >= 4096) return launch_w4<4, 0>(a, s);
    }
    if (g_variant == 2 || g_variant == 3 || g_variant == 4 || g_variant == 5 || g_variant == 6 || g_variant >= 0x20) return launch_pp<BM, BN, WM, WN, (BM + BN) / 64, 0, 0, 0>(a, s, LDS);
  }
  if (g_variant == 1 || (g_variant == 0 && a.act == VLB_ACT_SWIGLU_PAIR)) {
    return launch_pp<BM, BN, WM, WN, (BM + BN) / 64, 0, 0, 0>(a, s, LDS);
  }
  constexpr int T = (BM + BN) / 64;     // LDS-DMA instructions per thread per K-tile
  if constexpr (BM == 256 && BN == 256) {   // timing-only ablations of the 256x256 kernel (wrong results!)
    if (g_variant == 0x11) return launch_pp<BM, BN, WM, WN, T, 0, 0, 0, 1>(a, s, LDS);
    if (g_variant == 0x12) return launch_pp<BM, BN, WM, WN, T, 0, 0, 0, 2>(a, s, LDS);
    if (g_variant == 0x14) return launch_pp<BM, BN, WM, WN, T, 0, 0, 0, 4>(a, s, LDS);
    if (g_variant == 0x13) return launch_pp<BM, BN, WM, WN, T, 0, 0, 0, 3>(a, s, LDS);
    if (g_variant == 0x17) return launch_pp<BM, BN, WM, WN, T, 0, 0, 0, 7>(a, s, LDS);
    if (g_variant == 0x16) return launch_pp<BM, BN, WM, WN, T, 0, 0, 0, 6>(a, s, LDS);
  }
  if (g_variant >= 0x10) return launch_pp<BM, BN, WM, WN, T, 0, 0, 0>(a, s, LDS);   // ablations exist for 256x256 only
  if (g_variant != 0) {
    vlb_set_error("gemm: unknown kernel variant %d", g_variant);
    return VLB_ERR_INVALID;
  }
  hipLaunchKernelGGL((gemm_tile_kernel<BM, BN, WM, WN>), dim3(a.grid), dim3(512), LDS, s, a);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
#endif
}

// 16x16 LDS-tiled transpose, 32x32 elements per block
__global__ void transpose_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, int R, int C) {
  __shared__ bf16 t[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    if (r < R && c < C) t[i][threadIdx.x] = in[(int64_t)r * C + c];
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (r < R && c < C) out[(int64_t)c * R + r] = t[threadIdx.x][i];
  }
}

}  // namespace

// Tile choice exported for tests / DESIGN.md: 0 generic, 1 = 256x256, 2 = 256x128
extern "C" int vlb_gemm_kernel_choice(int M, int N, int K, int K2) {
  if (K % BK != 0 || K2 % BK != 0 || K == 0 || M < 128) return 0;
  const int cus = 256;
  if (N % 256 == 0) {
    const int t256 = ((M + 255) / 256) * (N / 256);
    // the 256x256 tile has the better MFMA:LDS ratio; fall back to 256x128 only when the grid
    // would leave more than half of the chip idle
    if (t256 < cus / 2) return 2;
    return 1;
  }
  if (N % 128 == 0) return 2;
  return 0;
}

// Tuning / test aid: how a long-K GEMM on the four-wave kernel is cut: rows*1000 + tail mode*100 + K splits
// (tail mode 0 whole tiles, 1 re-cut halves, 2 split-K).
extern "C" int vlb_gemm_plan(int M, int N, int K, int K2, int with_workspace) {
  if (N % 256 != 0 || K + K2 < 4096) return 0;
  TailPlan p256, p192;
  const bool use192 = plan_rows(M, N, K + K2, (with_workspace & 1) != 0, p256, p192, !(with_workspace & 2));     // bit 1: masked-pair rules
  const TailPlan& t = use192 ? p192 : p256;
  return (use192 ? 192 : 256) * 1000 + t.mode * 100 + t.splits;
}

#ifdef VLB_TOOLS
extern "C" int64_t vlb_gemm_workspace_bytes(void) { return SK_SLAB_BYTES + SK_FLAG_BYTES; }   // + the stream-K experiment's flags
#else
extern "C" int64_t vlb_gemm_workspace_bytes(void) { return 256ll * 256 * 256 * 4; }   // <= 256 work items x one 256x256 fp32 tile
#endif

extern "C" int vlb_gemm_bf16(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                             const void* bias, const void* residual, int ldr, int act, const void* A2, int lda2,
                             const void* W2, int ldw2, int K2, void* stream) {
  return vlb_gemm_bf16_ws(A, lda, W, ldw, C, ldc, M, N, K, bias, residual, ldr, act, A2, lda2, W2, ldw2, K2, nullptr, 0, stream);
}

static int gemm_impl(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                     const void* bias, const void* residual, int ldr, int act, const void* A2, int lda2,
                     const void* W2, int ldw2, int K2, void* ws, int64_t ws_bytes, void* stream, void* aux, int ldaux);

extern "C" int vlb_gemm_bf16_ws(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                                const void* bias, const void* residual, int ldr, int act, const void* A2, int lda2,
                                const void* W2, int ldw2, int K2, void* ws, int64_t ws_bytes, void* stream) {
  return gemm_impl(A, lda, W, ldw, C, ldc, M, N, K, bias, residual, ldr, act, A2, lda2, W2, ldw2, K2, ws, ws_bytes, stream, nullptr, 0);
}

extern "C" int vlb_gemm_swiglu_save(const void* A, int lda, const void* W_il, int ldw, void* H, int ldh, void* GU, int ldgu, int M, int N,
                                    int K, const void* A2, int lda2, const void* W2_il, int ldw2, int K2, void* ws, int64_t ws_bytes,
                                    void* stream) {
  VLB_REQUIRE(GU && ldgu >= N && ldgu % 4 == 0 && N % 8 == 0 && ((uintptr_t)GU % 8) == 0,
              "gemm_swiglu_save: [gate | up] rows must hold N columns, 8-byte aligned (N=%d ldgu=%d)", N, ldgu);
  return gemm_impl(A, lda, W_il, ldw, H, ldh, M, N, K, nullptr, nullptr, 0, VLB_ACT_SWIGLU_PAIR, A2, lda2, W2_il, ldw2, K2, ws, ws_bytes,
                   stream, GU, ldgu);
}

static int gemm_impl(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                     const void* bias, const void* residual, int ldr, int act, const void* A2, int lda2,
                     const void* W2, int ldw2, int K2, void* ws, int64_t ws_bytes, void* stream, void* aux, int ldaux) {
  VLB_REQUIRE(A && W && C, "gemm: null operand");
  VLB_REQUIRE(!ws || ((uintptr_t)ws % 16) == 0, "gemm: workspace must be 16-byte aligned");
  const bool ws_ok = ws && ws_bytes >= vlb_gemm_workspace_bytes();
  VLB_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
  VLB_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0, "gemm: K/lda/ldw must be multiples of 8 (K=%d lda=%d ldw=%d)", K, lda, ldw);
  VLB_REQUIRE(lda >= K && ldw >= K && (ldc >= N || (act == VLB_ACT_SWIGLU_PAIR && ldc >= N / 2)), "gemm: leading dimension smaller than row");
  if (K2 > 0) {
    VLB_REQUIRE(A2 && W2, "gemm: K2>0 needs A2 and W2");
    VLB_REQUIRE(K2 % 8 == 0 && lda2 % 8 == 0 && ldw2 % 8 == 0 && lda2 >= K2 && ldw2 >= K2, "gemm: bad second operand pair");
  } else {
    A2 = nullptr; W2 = nullptr; K2 = 0;
  }
  if (residual) VLB_REQUIRE(ldr >= N, "gemm: ldr < N");
  if (act == VLB_ACT_SWIGLU_PAIR) {
    VLB_REQUIRE(N % 32 == 0 && !bias && !residual && ldc >= N / 2, "gemm: SWIGLU_PAIR needs N %% 32 == 0, no bias/residual, ldc >= N/2");
  }
  GemmArgs a;
  a.A = (const bf16*)A; a.W = (const bf16*)W; a.C = (bf16*)C;
  a.A2 = (const bf16*)A2; a.W2 = (const bf16*)W2;
  a.bias = (const bf16*)bias; a.residual = (const bf16*)residual;
  a.M = M; a.N = N; a.K = K; a.K2 = K2;
  a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr; a.lda2 = lda2; a.ldw2 = ldw2;
  a.act = act; a.tiles_m = 0; a.tiles_n = 0; a.tile0 = 0; a.split_n = 1; a.grid = 0; a.drop_thresh = 0; a.drop_key = 0; a.drop_scale = 1.f;
  a.ws = nullptr; a.k_splits = 0; a.tail_tiles = 0; a.order = pick_order(M, N, K); a.aux = (bf16*)aux; a.ldaux = ldaux;
  a.stagger = g_stagger;
  a.wide = ((uintptr_t)C % 16) == 0 && ldc % 8 == 0 && (!aux || (((uintptr_t)aux % 16) == 0 && ldaux % 8 == 0 && (N / 2) % 8 == 0));
  hipStream_t s = as_stream(stream);
  const bool vec_ok = (ldc % 4 == 0) && (!residual || ldr % 4 == 0) && (!aux || (ldaux % 4 == 0 && (N / 2) % 4 == 0)) &&
                      (((uintptr_t)C | (uintptr_t)residual | (uintptr_t)bias | (uintptr_t)aux) % 8 == 0) &&
                      (((uintptr_t)A | (uintptr_t)W | (uintptr_t)A2 | (uintptr_t)W2) % 16 == 0);
  int choice = vec_ok ? vlb_gemm_kernel_choice(M, N, K, K2) : 0;
  if (choice != 0 && g_force_tile == 1 && N % 256 == 0) choice = 1;
  if (choice != 0 && g_force_tile == 2 && N % 128 == 0) choice = 2;
  if (choice == 1) {
    // Tail split: when the 256x256 grid ends in a mostly idle last wave of tiles (e.g. 2576 tiles = 10.06
    // waves for gate/up at M=5861), the full waves run as 256x256 tiles and the tiles of the partial wave
    // are re-cut into 256x128 halves in a second launch, which spreads them over twice as many CUs for
    // half as long.  Both launches walk the same tile order (map_tile), so the cut can be at any tile.
    const int cus = 256, tn = N / 256, tm = (M + 255) / 256;
    const int tiles = tm * tn, rem = tiles % cus;
    // Row quantisation: 192-row tiles (four-wave kernel only) when they cut the row count into fewer, fuller
    // waves - e.g. M=5861, N=4096: 23x16 = 368 tiles of 256 rows cost (1 + 0.62 tail) x 256 = 415 row-units,
    // 31x16 = 496 tiles of 192 rows cost 2 x 192 = 384.  Cost = waves x tile rows; a re-cut partial wave = 0.62.
    {
      const bool fits32 = a.wide && (int64_t)M * lda < (1ll << 31) && (int64_t)N * ldw < (1ll << 31) &&
                          (int64_t)M * lda2 < (1ll << 31) && (int64_t)N * ldw2 < (1ll << 31);
#ifdef VLB_TOOLS
      const int abl192 = (g_variant >= 0x30 && g_variant < 0x40) ? (g_variant & 0xf) : 0;     // timing-only ablations of the 192-row kernel
      const bool w4 = (g_variant == 3 || g_variant == 5 || g_variant == 6 || (g_variant >= 0x30 && g_variant < 0x40)) && fits32 && K + K2 >= 4096;    // 6 (A/B): never 192-row tiles
#else
      const bool w4 = (g_variant == 3 || g_variant == 5) && fits32 && K + K2 >= 4096;      // four-wave kernel shapes
#endif
#ifdef VLB_TOOLS
      // tools A/B: ragged multi-round output as ONE stream-K launch of 256-row tiles instead of rounds + a re-cut / split-K tail or 192-row tiles
      if (w4 && g_force_tile == 0 && g_variant == 3 && K % 64 == 0 && K2 % 64 == 0 && sk_wanted(tiles, (K + K2) / 64, act, ws_ok))
        return launch_w4_streamk<false>(a, tm, tn, (K + K2) / 64, ws, s);
#endif
      const int tm192 = (M + 191) / 192, tiles192 = tm192 * tn;
      TailPlan p256, p192;
      bool use192 = plan_rows(M, N, K + K2, ws_ok && w4, p256, p192);
#ifdef VLB_TOOLS
      if (g_variant == 6) use192 = false;
#endif
      if (w4 && g_force_tile == 0 && tiles > cus && (g_variant == 5 || (g_variant >= 0x30 && g_variant < 0x40) || use192)) {
        GemmArgs hi = a;
        hi.tiles_m = tm192; hi.tiles_n = tn; hi.tile0 = 0; hi.split_n = 1;
        const int rem192 = tiles192 % cus;
        hi.grid = p192.mode ? tiles192 - rem192 : tiles192;
#ifdef VLB_TOOLS
        if (abl192 == 1) return launch_w4<8, 1, 6>(hi, s);       // results wrong by construction: whole-tile launches only
        if (abl192 == 2) return launch_w4<8, 2, 6>(hi, s);
        if (abl192 == 4) return launch_w4<8, 4, 6>(hi, s);
        if (abl192 == 8) return launch_w4<8, 8, 6>(hi, s);
        if (abl192 == 7) return launch_w4<8, 7, 6>(hi, s);
        if (abl192 == 9) return launch_w4<8, 16, 6>(hi, s);      // 0x39: W always from panel 0
        if (abl192 == 10) return launch_w4<8, 32, 6>(hi, s);     // 0x3a: A always from panel 0
        if (abl192 == 11) return launch_w4<8, 48, 6>(hi, s);     // 0x3b: both
#endif
        int rc = launch_w4<8, 0, 6>(hi, s);
        if (rc != VLB_OK || !p192.mode) return rc;
        GemmArgs lo = hi;
        lo.tile0 = tiles192 - rem192;
        if (p192.mode == 2) {
          lo.ws = (float*)ws; lo.k_splits = p192.splits; lo.tail_tiles = rem192;
          return launch_w4_splitk<6>(lo, s);
        }
        lo.split_n = 2; lo.grid = 2 * rem192;
        return launch_w4<4, 0, 6>(lo, s);
      }
      if (w4 && g_force_tile == 0 && p256.mode == 2) {
        GemmArgs hi = a;
        hi.tiles_m = tm; hi.tiles_n = tn; hi.tile0 = 0; hi.split_n = 1; hi.grid = tiles - rem;
        int rc = launch_w4<8, 0, 8>(hi, s);
        if (rc != VLB_OK) return rc;
        GemmArgs lo = hi;
        lo.tile0 = tiles - rem; lo.ws = (float*)ws; lo.k_splits = p256.splits; lo.tail_tiles = rem;
        return launch_w4_splitk<8>(lo, s);
      }
    }
    if (g_tail_split && tiles > cus && rem != 0 && rem <= cus * 5 / 8) {
      GemmArgs hi = a, lo = a;
      hi.tiles_m = lo.tiles_m = tm;
      hi.tiles_n = lo.tiles_n = tn;
      hi.tile0 = 0; hi.split_n = 1; hi.grid = tiles - rem;
      lo.tile0 = tiles - rem; lo.split_n = 2; lo.grid = 2 * rem;
      int rc = launch_tile<256, 256, 2, 4>(hi, s);
      if (rc != VLB_OK) return rc;
      return launch_tile<256, 128, 4, 2>(lo, s);
    }
    return launch_tile<256, 256, 2, 4>(a, s);
  }
  if (choice == 2) return launch_tile<256, 128, 4, 2>(a, s);
  dim3 grid((N + 63) / 64, (M + 63) / 64);
  hipLaunchKernelGGL(gemm_generic_kernel, grid, dim3(256), 0, s, a);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

namespace {
inline uint32_t lowbias32_h(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
}  // namespace

extern "C" int vlb_gemm_bf16_masked_pair(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                                         const void* A2, int lda2, const void* W2, int ldw2, float drop_p, uint32_t seed,
                                         void* stream) {
  return vlb_gemm_bf16_masked_pair_ws(A, lda, W, ldw, C, ldc, M, N, K, A2, lda2, W2, ldw2, drop_p, seed, nullptr, 0, stream);
}

static int masked_pair_impl(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                            const void* A2, int lda2, const void* W2, int ldw2, float drop_p, uint32_t seed,
                            void* ws, int64_t ws_bytes, void* stream, int act, const void* residual, int ldr);

extern "C" int vlb_gemm_bf16_masked_pair_ws(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                                            const void* A2, int lda2, const void* W2, int ldw2, float drop_p, uint32_t seed,
                                            void* ws, int64_t ws_bytes, void* stream) {
  return masked_pair_impl(A, lda, W, ldw, C, ldc, M, N, K, A2, lda2, W2, ldw2, drop_p, seed, ws, ws_bytes, stream, VLB_ACT_NONE, nullptr, 0);
}

extern "C" int vlb_gemm_masked_pair_swiglu_bwd(const void* dY, int lddy, const void* Wt, int ldw, const void* gu, int ldgu, void* dgu,
                                               int lddgu, int M, int ff, int K, const void* U, int ldu, const void* At, int ldat,
                                               float drop_p, uint32_t seed, void* ws, int64_t ws_bytes, void* stream) {
  VLB_REQUIRE(gu && dgu, "gemm_masked_pair_swiglu_bwd: null activation pointers");
  VLB_REQUIRE(ldgu >= 2 * ff && lddgu >= 2 * ff && ldgu % 8 == 0 && lddgu % 8 == 0 && ff % 8 == 0 && ((uintptr_t)gu % 16) == 0 && ((uintptr_t)dgu % 16) == 0,
              "gemm_masked_pair_swiglu_bwd: [gate|up] rows must hold 2*ff columns; gu and d gu rows 16-byte aligned (ff=%d ldgu=%d lddgu=%d)", ff, ldgu, lddgu);
  return masked_pair_impl(dY, lddy, Wt, ldw, dgu, lddgu, M, ff, K, U, ldu, At, ldat, drop_p, seed, ws, ws_bytes, stream,
                          VLB_ACT_SWIGLU_BWD, gu, ldgu);
}

static int masked_pair_impl(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                            const void* A2, int lda2, const void* W2, int ldw2, float drop_p, uint32_t seed,
                            void* ws, int64_t ws_bytes, void* stream, int act, const void* residual, int ldr) {
  VLB_REQUIRE(!ws || ((uintptr_t)ws % 16) == 0, "gemm_masked_pair: workspace must be 16-byte aligned");
  const bool ws_ok = ws && ws_bytes >= vlb_gemm_workspace_bytes();
  VLB_REQUIRE(A && W && C && A2 && W2, "gemm_masked_pair: null operand");
  VLB_REQUIRE(M > 0 && N % 256 == 0 && K >= 128 && K % 64 == 0, "gemm_masked_pair: needs N %% 256 == 0, K %% 64 == 0, K >= 128 (N=%d K=%d)", N, K);
  VLB_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda2 % 8 == 0 && ldw2 % 8 == 0 && lda >= K && ldw >= K && lda2 >= 64 && ldw2 >= 64 &&
                  ldc % 4 == 0 && ldc >= N, "gemm_masked_pair: bad leading dimensions");
  VLB_REQUIRE((((uintptr_t)A | (uintptr_t)W | (uintptr_t)A2 | (uintptr_t)W2 | (uintptr_t)C) % 16) == 0 && ldc % 8 == 0,
              "gemm_masked_pair: operands and C rows must be 16-byte aligned");
  VLB_REQUIRE(drop_p > 0.f && drop_p < 1.f, "gemm_masked_pair: drop_p must be in (0,1) - use vlb_gemm_bf16 when p == 0");
  VLB_REQUIRE((int64_t)M * lda < (1ll << 31) && (int64_t)N * ldw < (1ll << 31) && (int64_t)M * lda2 < (1ll << 31) &&
                  (int64_t)N * ldw2 < (1ll << 31) && (int64_t)M * (N >> 1) < (1ll << 32),
              "gemm_masked_pair: operand too large for 32-bit offsets / the 32-bit dropout counter");
  GemmArgs a;
  a.A = (const bf16*)A; a.W = (const bf16*)W; a.C = (bf16*)C;
  a.A2 = (const bf16*)A2; a.W2 = (const bf16*)W2;
  a.bias = nullptr; a.residual = (const bf16*)residual;
  a.M = M; a.N = N; a.K = K; a.K2 = 64;
  a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr; a.lda2 = lda2; a.ldw2 = ldw2;
  a.act = act; a.tiles_m = 0; a.tiles_n = 0; a.tile0 = 0; a.split_n = 1; a.grid = 0;
  a.ws = nullptr; a.k_splits = 0; a.tail_tiles = 0; a.order = pick_order(M, N, K); a.aux = nullptr; a.ldaux = 0;
  a.wide = 1; a.stagger = g_stagger;
  uint32_t t = (uint32_t)(drop_p * 65536.f + 0.5f);
  a.drop_thresh = t > 65535u ? 65535u : t;
  a.drop_key = lowbias32_h(seed);
  a.drop_scale = 1.f / (1.f - drop_p);
  hipStream_t s = as_stream(stream);
  // same wave-quantisation choice as vlb_gemm_bf16 (256- or 192-row tiles; partial wave re-cut or split along K)
  const int cus = 256, tn = N / 256;
  TailPlan p256, p192;
  // (no split-K for the 256-row masked kernel: with 64 accumulator tiles + the mask temporaries the register
  // allocator starts rotating accumulator tuples behind the asm MFMAs; 192-row tiles are fine)
#ifdef VLB_TOOLS
  if (g_force_tile == 0 && sk_wanted(((M + 255) / 256) * tn, (K + 64) / 64, act, ws_ok))
    return launch_w4_streamk<true>(a, (M + 255) / 256, tn, (K + 64) / 64, ws, s);
#endif
  bool use192 = plan_rows(M, N, K + 64, ws_ok, p256, p192, false);
#ifdef VLB_TOOLS
  if (g_force_tile == 3) use192 = false;      // A/B only: force 256- / 192-row tiles for the masked-pair kernel
  if (g_force_tile == 4) use192 = true;
#endif
  const TailPlan& tp = use192 ? p192 : p256;
  const int tm = use192 ? (M + 191) / 192 : (M + 255) / 256;
  const int tiles = tm * tn, rem = tiles % cus;
  GemmArgs hi = a;
  hi.tiles_m = tm; hi.tiles_n = tn; hi.tile0 = 0; hi.split_n = 1;
  hi.grid = tp.mode ? tiles - rem : tiles;
  int rc = use192 ? launch_w4<8, 0, 6, true>(hi, s) : launch_w4<8, 0, 8, true>(hi, s);
  if (rc != VLB_OK || !tp.mode) return rc;
  GemmArgs lo = hi;
  lo.tile0 = tiles - rem;
  if (tp.mode == 2) {
    lo.ws = (float*)ws; lo.k_splits = tp.splits; lo.tail_tiles = rem;
    return launch_w4_splitk<6, true>(lo, s);
  }
  lo.split_n = 2; lo.grid = 2 * rem;
  return use192 ? launch_w4<4, 0, 6, true>(lo, s) : launch_w4<4, 0, 8, true>(lo, s);
}

#ifdef VLB_TOOLS
// tuning hooks, libvlb_tools.so only: kernel variant / forced tile
extern "C" void vlb_gemm_set_stagger(int ticks) { g_stagger = ticks; }
extern "C" void vlb_gemm_set_streamk(int on) { g_streamk = on; }
extern "C" void vlb_gemm_set_rowsplit(int on) { g_w4_rowsplit = on; }
extern "C" void vlb_gemm_set_persist(int on) { g_w4_persist = on; }
extern "C" void vlb_gemm_set_wd(int on) { g_wd = on; }
extern "C" void vlb_gemm_set_variant(int variant, int force_tile) {
  g_variant = variant & 0xff;     // 3 = default (auto)
  g_force_tile = force_tile;
  g_tail_split = (variant & 0x100) ? 0 : 1;    // bit 8 disables the tail split (A/B)
  g_tail_splitk = (variant & 0x200) ? 0 : 1;   // bit 9 disables the split-K tail (A/B)
  g_order_auto = (variant & 0x2000) ? 0 : 1;
  g_tile_order = ((variant >> 10) & 7) ^ 3;       // variant word 3 = product default (order 3)
}
#endif

extern "C" int vlb_transpose_bf16(const void* in, void* out, int R, int C, void* stream) {
  VLB_REQUIRE(in && out && R > 0 && C > 0, "transpose: bad args");
  dim3 grid((C + 31) / 32, (R + 31) / 32), block(32, 8);
  hipLaunchKernelGGL(transpose_kernel, grid, block, 0, as_stream(stream), (const bf16*)in, (bf16*)out, R, C);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
