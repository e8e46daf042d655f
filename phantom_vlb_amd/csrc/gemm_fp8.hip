// Block-scaled fp8 (MX: OCP e4m3 elements, one E8M0 scale per 32 consecutive K elements) MFMA GEMM for gfx950:
//   C[M,N] (bf16) = dequant(Aq, sA)[M,K] . dequant(Wq, sW)[N,K]^T (+ residual)
// on v_mfma_scale_f32_16x16x128_f8f6f4 - the instruction that reaches the fp8 rate (2x the bf16 MFMA rate per clock;
// the unscaled fp8 MFMAs run at the bf16 rate, MI355X_MICROARCH.md "Matrix cores").  BASELINE configs[4] asks for
// fp8 MFMA GEMMs on the full fine-tune's linears; vlb_quantize_mxfp8 produces the operands from bf16.
//
// Kernel: 256x256 output tile, K step 128 (= one MFMA K), 256 threads = 2(M) x 2(N) waves of 128x128; operands go
// global -> LDS by LDS-DMA (global_load_lds_dwordx4) into a double-buffered image of 128-byte rows whose 16-byte
// chunks are XOR-swizzled on the SOURCE address (chunk c of row r at slot c ^ ((r>>1)&7)).  Operand layout of the
// instruction, measured with exact data (tools/probe_mfma_scale.py): lane l = (row l&15, group g = l>>4) holds
// k = 16g..16g+15 in its first 16 bytes and k = 64+16g..64+16g+15 in its second 16 bytes (two ds_read_b128: chunks g
// and 4+g of the row), while the E8M0 scale it supplies is the one of 32-element block g (k = 32g..32g+31) of its
// row; both operands' scales are bytes of ONE register (see the K loop).  MFMA roles are swapped like in the
// bf16 kernel (MFMA rows = output columns) so a lane owns 4 consecutive n of one output row: 8-byte stores.
// Scales: one byte per (row, 32-element K block); the u32 of a row's four blocks per K-tile rides the same LDS-DMA ring.
#include <type_traits>

#include "common.hpp"

namespace {
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct Fp8Args {
  const uint8_t* A; const uint8_t* sA; const uint8_t* W; const uint8_t* sW;
  bf16* C; const bf16* residual;
  int M, N, K, lda, ldw, ldc, ldr, ldsa, ldsw, tiles_m, tiles_n;
};

constexpr int FBK = 128;            // K bytes per tile row
constexpr int FROW = 128;           // LDS row bytes
__device__ __forceinline__ int f_off(int r, int c) { return r * FROW + ((c ^ ((r >> 1) & 7)) << 4); }
__device__ __forceinline__ void glds16f(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g, (void __attribute__((address_space(3)))*)l, 16, 0, 0);
}

__device__ __forceinline__ void glds4f(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g, (void __attribute__((address_space(3)))*)l, 4, 0, 0);
}

struct Fp8Launch { int tile0, split_n; };       // first tile id of this launch; 1 = whole tiles, 2 = column halves
#ifdef VLB_TOOLS          // the read-phase kernel of rounds 1-2, kept in the tools build for A/B (tools/ab_fp8_variants.py)
// 256 threads = 2(M) x 2(N) waves, one per SIMD with the full 512-register budget; wave block 128 x 128 = 64
// accumulator tiles.  Per K-tile (128 bytes of K per row) a wave reads all its fragments (32 ds_read_b128) and scale
// words into registers, the workgroup passes a barrier, and the stage just read is immediately refilled with tile
// kt+2 by LDS-DMA while the 64 MFMAs of tile kt run - every DMA has one whole iteration plus an MFMA phase to land
// (counted vmcnt, never 0 inside the loop).  Scales travel through LDS as well (global_load_lds_dword, one u32 = the
// four block scales of a row per K-tile), so the loop holds no register-destination global load for the compiler to
// drain the DMA queue on.
// NT = 8: 256 x 256 tile (wave block 128 x 128).  NT = 4: 256 x 128 tile (wave block 128 x 64) - the partial last
// wave of tiles of a GEMM is re-cut into these halves so its work spreads over twice as many CUs.

template <int NT>
__global__ __launch_bounds__(256, 1) void gemm_mxfp8_kernel(Fp8Args p, Fp8Launch L) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int A_BYTES = 256 * FROW, WROWS = 32 * NT, W_BYTES = WROWS * FROW;
  constexpr int SA_OFF = A_BYTES + W_BYTES, SW_OFF = SA_OFF + 256 * 4;
  constexpr int STAGE = SW_OFF + 256 * 4;                    // A image + W image + A scale words + W scale words
  constexpr int VM = 8 + NT + 2;                             // LDS-DMA instructions per wave per K-tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // ---- work item -> tile (same order as gemm.hip, tile order 3): blocks b and b+8 share an XCD; every round of 256 work
  // items is dealt to the XCDs in chunks of 32 (a partial last round in contiguous runs), and tiles are walked in column
  // bands of 8 with the rows inside a band - a chunk is 4 rows x 8 columns, a chip-wide round sweeps the row panels of A
  // against the same 8 W panels, which the XCDs then share through the Infinity Cache
  int w;
  {
    const int n = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int T = n >> 8, tr = idx >> 5;
    if (tr < T) {
      w = (tr << 8) + (xcd << 5) + (idx & 31);
    } else {
      const int base = T << 8, nn = n - base, q = nn >> 3, r = nn & 7;
      w = base + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (idx - (T << 5));
    }
  }
  const int t = L.tile0 + w / L.split_n, half = w % L.split_n;
  const int band = t / (8 * p.tiles_m), c0 = band * 8, within = t - band * 8 * p.tiles_m;
  const int band_cols = min(8, p.tiles_n - c0);
  const int tm = within / band_cols, tn = c0 + within % band_cols;
  const int m0 = tm * 256, n0 = tn * 256 + half * 128;
  const int nk = p.K / FBK;

  // ---- LDS-DMA sources.  Data: a wave instruction writes 1 KiB = 8 rows x 128 B; lane -> (row, slot); source chunk =
  // slot ^ swz(row).  Wave w owns A rows 64w..64w+63 (8 pieces) and W rows 8*NT*w.. (NT pieces).  Scales: lane l of
  // wave w moves the u32 of A row 64w + l and of W row (64w + l) mod WROWS (for NT = 4 the upper waves repeat rows
  // into an unused part of the scale area: every wave issues the same number of DMA instructions).
  const char* srcA[8]; const char* srcW[NT];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = wave * 64 + j * 8 + (lane >> 3), s = lane & 7;
    srcA[j] = reinterpret_cast<const char*>(p.A) + (int64_t)min(m0 + r, p.M - 1) * p.lda + (s ^ ((r >> 1) & 7)) * 16;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int r = wave * 8 * NT + j * 8 + (lane >> 3), s = lane & 7;
    srcW[j] = reinterpret_cast<const char*>(p.W) + (int64_t)(n0 + r) * p.ldw + (s ^ ((r >> 1) & 7)) * 16;
  }
  const char* srcSA = reinterpret_cast<const char*>(p.sA) + (int64_t)min(m0 + wave * 64 + lane, p.M - 1) * p.ldsa;
  const char* srcSW = reinterpret_cast<const char*>(p.sW) + (int64_t)(n0 + (wave * 64 + lane) % WROWS) * p.ldsw;
  auto stage = [&](int st, int kt) {
    char* base = smem + st * STAGE;
#pragma unroll
    for (int j = 0; j < 8; ++j) glds16f(srcA[j] + (int64_t)kt * FBK, base + (wave * 8 + j) * 1024);
#pragma unroll
    for (int j = 0; j < NT; ++j) glds16f(srcW[j] + (int64_t)kt * FBK, base + A_BYTES + (wave * NT + j) * 1024);
    glds4f(srcSA + kt * 4, base + SA_OFF + wave * 256);
    glds4f(srcSW + kt * 4, base + SW_OFF + wave * 256);
  };
  const int fr = lane & 15, g = lane >> 4;
  f32x4 acc[NT][8];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

#define VLB_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
  stage(0, 0);
  if (nk > 1) stage(1, 1);
  if (nk > 1) VLB_VMCNT(VM); else VLB_VMCNT(0);
  __builtin_amdgcn_s_barrier();
  // One K-tile.  MORE: tile kt+2 exists and is issued into the stage this tile was read from.  After the barrier the
  // LDS-DMA pieces of tile kt+2 are issued two or three at a time in front of each row of NT MFMAs (sched_barrier fences
  // pin that order), so their issue cost (60-100 cycles apiece) hides in the shadow of the previous row's MFMAs instead
  // of standing as one burst in front of the whole MFMA phase.
  auto tile = [&](int kt, auto more) {
    constexpr bool MORE = decltype(more)::value;
    const char* sb = smem + (kt & 1) * STAGE;
    char* nb = smem + (kt & 1) * STAGE;                    // stage refilled with tile kt+2
    i32x8 wf[NT], af[8];
    int swb[NT], sab[8];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int r = wn * 16 * NT + j * 16 + fr;
      const i32x4 lo = *reinterpret_cast<const i32x4*>(sb + A_BYTES + f_off(r, g));
      const i32x4 hi = *reinterpret_cast<const i32x4*>(sb + A_BYTES + f_off(r, 4 + g));
      wf[j] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      swb[j] = *reinterpret_cast<const int*>(sb + SW_OFF + r * 4);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = wm * 128 + i * 16 + fr;
      const i32x4 lo = *reinterpret_cast<const i32x4*>(sb + f_off(r, g));
      const i32x4 hi = *reinterpret_cast<const i32x4*>(sb + f_off(r, 4 + g));
      af[i] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      sab[i] = *reinterpret_cast<const int*>(sb + SA_OFF + r * 4);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                          // every wave holds tile kt in registers: its stage is free
    // Both E8M0 scales travel in ONE register: byte 0 = the MFMA-A operand's (the W rows), byte 1 = the MFMA-B
    // operand's (the activation rows), selected by op_sel 0 / 1.  Measured on gfx950 with ROCm 7.2
    // (tools/probe_mfma_scale.py): the instruction takes both scale bytes from the register in the scale_b position
    // and ignores the scale_a register; passing the same packed register in both positions is right under either reading.
#pragma unroll
    for (int j = 0; j < NT; ++j) swb[j] = (swb[j] >> (8 * g)) & 0xff;
#pragma unroll
    for (int i = 0; i < 8; ++i) sab[i] = ((sab[i] >> (8 * g)) & 0xff) << 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (MORE) {
        const int64_t ko = (int64_t)(kt + 2) * FBK;
        glds16f(srcA[i] + ko, nb + (wave * 8 + i) * 1024);
        if (i < NT) glds16f(srcW[i] + ko, nb + A_BYTES + (wave * NT + i) * 1024);
        if (i == 0) glds4f(srcSA + (kt + 2) * 4, nb + SA_OFF + wave * 256);
        if (i == 1) glds4f(srcSW + (kt + 2) * 4, nb + SW_OFF + wave * 256);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int sc = swb[j] | sab[i];
        acc[j][i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[j], af[i], acc[j][i], 0, 0, 0, sc, 1, sc);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // tile kt+1 must have landed before the next iteration reads it: everything older than tile kt+2's pieces
    if constexpr (MORE) VLB_VMCNT(VM); else VLB_VMCNT(0);
    __builtin_amdgcn_s_barrier();
  };
  int kt = 0;
  for (; kt + 2 < nk; ++kt) tile(kt, std::true_type{});
  for (; kt < nk; ++kt) tile(kt, std::false_type{});
#undef VLB_VMCNT
  // ---- epilogue: lane holds, for tile (j, i): output row m = .. + fr, columns n = .. + 4g + {0,1,2,3}
  // (C rows 16-byte aligned: adjacent fragments paired into 16-byte stores, store_pair16 in common.hpp; else 8-byte pieces)
  auto value = [&](const f32x4& a, int m, int n) {
    f32x4 v = a;
    if (p.residual) {
      const bf16x4 r = *reinterpret_cast<const bf16x4*>(p.residual + (int64_t)m * p.ldr + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
    }
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
    return o;
  };
  const bool wide = ((uintptr_t)p.C % 16) == 0 && p.ldc % 8 == 0;
  if (wide) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + wm * 128 + i * 16 + fr;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NT; j += 2) {
        const int n = n0 + wn * 16 * NT + j * 16 + 4 * g;
        store_pair16(p.C + (int64_t)m * p.ldc, n, value(acc[j][i], m, n), value(acc[j + 1][i], m, n + 16), g);
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wm * 128 + i * 16 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wn * 16 * NT + j * 16 + 4 * g;
      *reinterpret_cast<bf16x4*>(p.C + (int64_t)m * p.ldc + n) = value(acc[j][i], m, n);
    }
  }
}
#endif

// ---------------------------------------------------------------- pipelined variant (round 3)
// Same tile, LDS image and operand layout as gemm_mxfp8_kernel, but no fragment-read phase in front of the MFMAs: the 64
// MFMAs of a K-tile run as two blocks of NT groups, group j = W fragment j against activation rows 0..3 (block 1) or 4..7
// (block 2), and every ds_read is issued under MFMAs that do not need it:
//   block 1 of tile kt: reads activation fragments 4..7 of tile kt; after barrier S (every wave holds all of W(kt)) the W
//                       image of this stage is refilled with W(kt+2) by LDS-DMA
//   barrier M:          every wave holds all of tile kt, and tile kt+1 has landed (counted vmcnt: only W(kt+2) may fly)
//   block 2 of tile kt: LDS-DMA of A(kt+2) into this stage's A image; reads of tile kt+1 from the other stage: activation
//                       fragments 0..3 up front (their registers died with block 1), W fragment j IN PLACE right behind the
//                       last MFMA group that uses it
// so the register file holds one set of fragments (8 + NT), the DMA pieces are spread over both blocks, and each piece has
// one to two blocks (1000-3000 matrix cycles) to land.  Scale bytes are fetched with ds_read_u8 (byte g of the row's word).
// MT = 8: 256-row tiles (wave block 128 rows); MT = 6: 192-row tiles (wave block 96 rows), picked by the host when they
// quantise the row count into fewer, fuller rounds of tiles (plan_fp8_rows).
template <int NT, int S_AT, int MT = 8>
__global__ __launch_bounds__(256, 1) void gemm_mxfp8_pipe_kernel(Fp8Args p, Fp8Launch L) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int AROWS = 32 * MT, HM = MT / 2;
  constexpr int A_BYTES = AROWS * FROW, WROWS = 32 * NT, W_BYTES = WROWS * FROW;
  constexpr int SA_OFF = A_BYTES + W_BYTES, SW_OFF = SA_OFF + 256 * 4;
  constexpr int STAGE = SW_OFF + 256 * 4;
  constexpr int NW = NT + 1, NA = MT + 1;                       // LDS-DMA pieces per wave per K-tile: W image + scale words, A image + scale words
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int w;
  {
    const int n = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int T = n >> 8, tr = idx >> 5;
    if (tr < T) {
      w = (tr << 8) + (xcd << 5) + (idx & 31);
    } else {
      const int base = T << 8, nn = n - base, q = nn >> 3, r = nn & 7;
      w = base + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (idx - (T << 5));
    }
  }
  const int t = L.tile0 + w / L.split_n, half = w % L.split_n;
  const int band = t / (8 * p.tiles_m), c0 = band * 8, within = t - band * 8 * p.tiles_m;
  const int band_cols = min(8, p.tiles_n - c0);
  const int tm = within / band_cols, tn = c0 + within % band_cols;
  const int m0 = tm * AROWS, n0 = tn * 256 + half * 128;
  const int nk = p.K / FBK;

  // LDS-DMA sources = uniform base (advanced 128 B per K-tile) + one 32-bit byte offset per piece (host checks the extents)
  uint32_t offA[MT], offW[NT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int r = wave * 8 * MT + j * 8 + (lane >> 3), s = lane & 7;
    offA[j] = (uint32_t)min(m0 + r, p.M - 1) * (uint32_t)p.lda + (uint32_t)((s ^ ((r >> 1) & 7)) * 16);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int r = wave * 8 * NT + j * 8 + (lane >> 3), s = lane & 7;
    offW[j] = (uint32_t)(n0 + r) * (uint32_t)p.ldw + (uint32_t)((s ^ ((r >> 1) & 7)) * 16);
  }
  const uint32_t offSA = (uint32_t)min(m0 + (wave * 64 + lane) % AROWS, p.M - 1) * (uint32_t)p.ldsa;
  const uint32_t offSW = (uint32_t)(n0 + (wave * 64 + lane) % WROWS) * (uint32_t)p.ldsw;
  const char* const baseA = reinterpret_cast<const char*>(p.A); const char* const baseW = reinterpret_cast<const char*>(p.W);
  const char* const baseSA = reinterpret_cast<const char*>(p.sA); const char* const baseSW = reinterpret_cast<const char*>(p.sW);
  auto dmaW = [&](int st, int kt, int pc) {                  // piece pc of W(kt): 0..NT-1 image, NT scale words
    char* base = smem + st * STAGE;
    if (pc < NT) glds16f(baseW + (int64_t)kt * FBK + offW[pc], base + A_BYTES + (wave * NT + pc) * 1024);
    else glds4f(baseSW + kt * 4 + offSW, base + SW_OFF + wave * 256);
  };
  auto dmaA = [&](int st, int kt, int pc) {                  // piece pc of A(kt): 0..MT-1 image, MT scale words
    char* base = smem + st * STAGE;
    if (pc < MT) glds16f(baseA + (int64_t)kt * FBK + offA[pc], base + (wave * MT + pc) * 1024);
    else glds4f(baseSA + kt * 4 + offSA, base + SA_OFF + wave * 256);
  };
  const int fr = lane & 15, g = lane >> 4;
  f32x4 acc[NT][MT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  i32x8 wf[NT], af[MT];
  int swb[NT], sab[MT];
  auto readW = [&](const char* sb, int j) {
    const int r = wn * 16 * NT + j * 16 + fr;
    const i32x4 lo = *reinterpret_cast<const i32x4*>(sb + A_BYTES + f_off(r, g));
    const i32x4 hi = *reinterpret_cast<const i32x4*>(sb + A_BYTES + f_off(r, 4 + g));
    wf[j] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    swb[j] = *reinterpret_cast<const uint8_t*>(sb + SW_OFF + r * 4 + g);
  };
  auto readA = [&](const char* sb, int i) {
    const int r = wm * 16 * MT + i * 16 + fr;
    const i32x4 lo = *reinterpret_cast<const i32x4*>(sb + f_off(r, g));
    const i32x4 hi = *reinterpret_cast<const i32x4*>(sb + f_off(r, 4 + g));
    af[i] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    sab[i] = *reinterpret_cast<const uint8_t*>(sb + SA_OFF + r * 4 + g);
  };
#define VLB_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define VLB_FENCE() __builtin_amdgcn_sched_barrier(0)
#pragma unroll
  for (int pc = 0; pc < NW; ++pc) dmaW(0, 0, pc);
#pragma unroll
  for (int pc = 0; pc < NA; ++pc) dmaA(0, 0, pc);
  if (nk > 1) {
#pragma unroll
    for (int pc = 0; pc < NW; ++pc) dmaW(1, 1, pc);
#pragma unroll
    for (int pc = 0; pc < NA; ++pc) dmaA(1, 1, pc);
    VLB_VMCNT(NW + NA);
  } else {
    VLB_VMCNT(0);
  }
  VLB_FENCE(); __builtin_amdgcn_s_barrier(); VLB_FENCE();
#pragma unroll
  for (int j = 0; j < NT - 1; ++j) readW(smem, j);           // (fragment NT-1 is fetched at the top of every tile)
#pragma unroll
  for (int i = 0; i < HM; ++i) readA(smem, i);
  __builtin_amdgcn_s_waitcnt(0xc07f);                        // so that no compiler-inserted full wait lands at the loop header
  VLB_FENCE();

  // MORE: a tile kt+1 exists (its fragments are fetched in block 2); LOAD2: a tile kt+2 exists (it is DMAed into this stage)
  auto tile = [&](int kt, auto more_c, auto load2_c) {
    constexpr bool MORE = decltype(more_c)::value, LOAD2 = decltype(load2_c)::value;
    const int st = kt & 1;
    const char* sb = smem + st * STAGE;
    const char* sn = smem + (st ^ 1) * STAGE;
    // ---------------- block 1: W fragment j x activation rows 0..3 || reads of rows 4..7 || DMA of W(kt+2)
    readW(sb, NT - 1);                                       // the one W fragment block 2 of the previous tile could not refresh in time
#pragma unroll
    for (int i = HM; i < MT; ++i) readA(sb, i);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if constexpr (LOAD2) {
        if (j == S_AT) {                                     // S: every wave holds W(kt) - this stage's W image is free
          __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0), as an instruction the compiler's own wait insertion accounts for
          VLB_FENCE(); __builtin_amdgcn_s_barrier(); VLB_FENCE();
        }
        if (j >= S_AT) {
#pragma unroll
          for (int pc = 0; pc < NW; ++pc)
            if (pc * (NT - S_AT) / NW == j - S_AT) dmaW(st, kt + 2, pc);
        }
      }
#pragma unroll
      for (int i = 0; i < HM; ++i) {
        const int sc = swb[j] | (sab[i] << 8);
        acc[j][i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[j], af[i], acc[j][i], 0, 0, 0, sc, 1, sc);
      }
      VLB_FENCE();
    }
    if constexpr (MORE) {
      // M: every wave holds all of tile kt (its A image is free); tile kt+1 has landed - only W(kt+2)'s pieces may still fly
      __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0), as an instruction the compiler's own wait insertion accounts for
      if constexpr (LOAD2) VLB_VMCNT(NW); else VLB_VMCNT(0);
      VLB_FENCE(); __builtin_amdgcn_s_barrier(); VLB_FENCE();
    }
    // ---------------- block 2: W fragment j x activation rows 4..7 || DMA of A(kt+2) || reads of tile kt+1
    if constexpr (MORE) {
#pragma unroll
      for (int i = 0; i < HM; ++i) readA(sn, i);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if constexpr (LOAD2) {
#pragma unroll
        for (int pc = 0; pc < NA; ++pc)
          if (pc * NT / NA == j) dmaA(st, kt + 2, pc);
      }
      if constexpr (MORE) { if (j >= 1) readW(sn, j - 1); }  // in place: fragment j-1 died with the previous group
#pragma unroll
      for (int i = HM; i < MT; ++i) {
        const int sc = swb[j] | (sab[i] << 8);
        acc[j][i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[j], af[i], acc[j][i], 0, 0, 0, sc, 1, sc);
      }
      VLB_FENCE();
    }
  };
  {
    using T_ = std::true_type; using F_ = std::false_type;
    int kt = 0;
    for (; kt + 2 < nk; ++kt) tile(kt, T_{}, T_{});
    if (kt + 1 < nk) { tile(kt, T_{}, F_{}); ++kt; }
    tile(kt, F_{}, F_{});
  }
#undef VLB_VMCNT
#undef VLB_FENCE
  // ---- epilogue: lane holds, for tile (j, i): output row m = .. + fr, columns n = .. + 4g + {0,1,2,3}.  One fully
  // unrolled loop nest per (residual?, 16-byte rows?) case, chosen once (compact instruction stream, no per-fragment branch).
  auto cvt = [](const f32x4& v) {
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
    return o;
  };
  auto rows = [&](auto res_c, auto wide_c) {
    constexpr bool RES = decltype(res_c)::value, WIDE = decltype(wide_c)::value;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + wm * 16 * MT + i * 16 + fr;
      if (m >= p.M) continue;
      bf16* crow = p.C + (int64_t)m * p.ldc;
#pragma unroll
      for (int j = 0; j < NT; j += 2) {
        const int n = n0 + wn * 16 * NT + j * 16 + 4 * g;
        f32x4 va = acc[j][i], vb = acc[j + 1][i];
        if constexpr (RES) {
          const bf16* rp = p.residual + (int64_t)m * p.ldr + n;
          const bf16x4 ra = *reinterpret_cast<const bf16x4*>(rp), rb = *reinterpret_cast<const bf16x4*>(rp + 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) { va[e] += (float)ra[e]; vb[e] += (float)rb[e]; }
        }
        if constexpr (WIDE) {
          store_pair16(crow, n, cvt(va), cvt(vb), g);
        } else {
          *reinterpret_cast<bf16x4*>(crow + n) = cvt(va);
          *reinterpret_cast<bf16x4*>(crow + n + 16) = cvt(vb);
        }
      }
    }
  };
  using ET_ = std::true_type; using EF_ = std::false_type;
  const bool wide = ((uintptr_t)p.C % 16) == 0 && p.ldc % 8 == 0;
  if (wide) { if (p.residual) rows(ET_{}, ET_{}); else rows(EF_{}, ET_{}); }
  else { if (p.residual) rows(ET_{}, EF_{}); else rows(EF_{}, EF_{}); }
}

#ifdef VLB_TOOLS
int g_fp8_rows = 0;                 // tools: 0 = planner, 256 / 192 = forced tile height
int g_fp8_variant = 1;              // tools: 0 = read-phase kernel (rounds 1-2), 1 = pipelined kernel
#else
constexpr int g_fp8_variant = 1;
#endif

template <int NT, int MT = 8>
int launch_fp8(const Fp8Args& a, Fp8Launch L, int grid, hipStream_t st) {
  constexpr int LDS = 2 * (32 * MT * FROW + 32 * NT * FROW + 2 * 256 * 4);
  if constexpr (MT != 8) {
    static const hipError_t attr6 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mxfp8_pipe_kernel<NT, 1, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (attr6 != hipSuccess) { vlb_set_error("gemm_mxfp8: cannot reserve %d bytes of LDS: %s", LDS, hipGetErrorString(attr6)); return VLB_ERR_LAUNCH; }
    hipLaunchKernelGGL((gemm_mxfp8_pipe_kernel<NT, 1, MT>), dim3(grid), dim3(256), LDS, st, a, L);
    VLB_LAUNCH_CHECK();
    return VLB_OK;
  }
#ifdef VLB_TOOLS
  if (g_fp8_variant == 2 || g_fp8_variant == 3) {            // tools: barrier S in front of MFMA group 2 / 0 instead of 1
    auto k = g_fp8_variant == 2 ? &gemm_mxfp8_pipe_kernel<NT, 2> : &gemm_mxfp8_pipe_kernel<NT, 0>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return VLB_ERR_LAUNCH;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), LDS, st, a, L);
    VLB_LAUNCH_CHECK();
    return VLB_OK;
  }
#endif
  if (g_fp8_variant == 1) {
    static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mxfp8_pipe_kernel<NT, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (attr1 != hipSuccess) { vlb_set_error("gemm_mxfp8: cannot reserve %d bytes of LDS: %s", LDS, hipGetErrorString(attr1)); return VLB_ERR_LAUNCH; }
    hipLaunchKernelGGL((gemm_mxfp8_pipe_kernel<NT, 1>), dim3(grid), dim3(256), LDS, st, a, L);
    VLB_LAUNCH_CHECK();
    return VLB_OK;
  }
#ifdef VLB_TOOLS
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mxfp8_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  if (attr != hipSuccess) { vlb_set_error("gemm_mxfp8: cannot reserve %d bytes of LDS: %s", LDS, hipGetErrorString(attr)); return VLB_ERR_LAUNCH; }
  hipLaunchKernelGGL((gemm_mxfp8_kernel<NT>), dim3(grid), dim3(256), LDS, st, a, L);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
#else
  return VLB_ERR_LAUNCH;
#endif
}

// ---------------------------------------------------------------- bf16 -> MX fp8 (e4m3 + E8M0 block scales)
// One thread per 32-element block: amax -> shared exponent E = ceil(log2(amax / 448)) clamped to [-127, 127];
// scale byte = E + 127; elements = RNE(x * 2^-E) saturated to +-448 (OCP e4m3fn: no infinities).
__device__ __forceinline__ uint8_t f32_to_e4m3(float x) {
  // round-to-nearest-even into e4m3fn (bias 7, 3 mantissa bits, max 448, subnormal step 2^-9)
  const uint32_t sign = (__float_as_uint(x) >> 24) & 0x80u;
  float a = fabsf(x);
  if (!(a == a)) return (uint8_t)(sign | 0x7f);             // NaN
  if (a >= 448.f) return (uint8_t)(sign | 0x7e);            // saturate to the largest finite value
  if (a < 0.0009765625f) return (uint8_t)sign;              // below half the smallest subnormal (2^-10): zero
  int e; (void)frexpf(a, &e);                               // a = f * 2^e, f in [0.5, 1)
  int exp_unb = e - 1;                                      // a = 1.m * 2^exp_unb
  if (exp_unb < -6) exp_unb = -6;                           // subnormal range shares the exponent of the smallest normal
  const float q = rintf(a * exp2f((float)(3 - exp_unb)));   // mantissa steps of 2^(exp_unb-3), RNE (rintf)
  float r = q * exp2f((float)(exp_unb - 3));
  if (r >= 448.f) return (uint8_t)(sign | 0x7e);
  if (r == 0.f) return (uint8_t)sign;
  (void)frexpf(r, &e);
  int eu = e - 1;
  uint32_t bits;
  if (eu < -6) bits = (uint32_t)rintf(r * 512.f);           // subnormal: value = m * 2^-9
  else bits = ((uint32_t)(eu + 7) << 3) | ((uint32_t)rintf(r * exp2f((float)(3 - eu))) & 7u);
  return (uint8_t)(sign | bits);
}
__global__ __launch_bounds__(256) void quantize_mxfp8_kernel(const bf16* __restrict__ x, int ldx, uint8_t* __restrict__ q, int ldq,
                                                            uint8_t* __restrict__ s, int lds_, int rows, int kblocks) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * kblocks) return;
  const int r = (int)(i / kblocks), kb = (int)(i % kblocks);
  const bf16* xp = x + (int64_t)r * ldx + kb * 32;
  float v[32];
  float amax = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(xp + c * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[c * 8 + k] = (float)t[k]; amax = fmaxf(amax, fabsf(v[c * 8 + k])); }
  }
  int E = -127;
  if (amax > 0.f) {
    int e; const float f = frexpf(amax / 448.f, &e);        // amax/448 = f * 2^e, f in [0.5,1)
    E = (f == 0.5f) ? e - 1 : e;                            // ceil(log2(amax/448))
    E = max(-127, min(127, E));
  }
  s[(int64_t)r * lds_ + kb] = (uint8_t)(E + 127);
  const float inv = exp2f((float)-E);
  // v_cvt_pk_fp8_f32: two floats -> two OCP e4m3 bytes, round-to-nearest-even (the scaled block never exceeds 448)
  uint32_t w[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    int pk = 0;
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[c * 4] * inv, v[c * 4 + 1] * inv, pk, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[c * 4 + 2] * inv, v[c * 4 + 3] * inv, pk, true);
    w[c] = (uint32_t)pk;
  }
  u32x4* qp = reinterpret_cast<u32x4*>(q + (int64_t)r * ldq + kb * 32);
  qp[0] = u32x4{w[0], w[1], w[2], w[3]};
  qp[1] = u32x4{w[4], w[5], w[6], w[7]};
}
// ---------------------------------------------------------------- transpose + quantise in one pass
// q[C, Rpad] (e4m3) and s[C, Rpad/32] (E8M0) of x[R, C]^T: what the dgrad (W^T) and wgrad (dy^T, x^T) GEMMs of the fp8
// path consume - MX blocks run along the ORIGINAL row axis.  Tile 128 (R) x 128 (C) through LDS (row stride 65 words:
// column walks are conflict-free): 16-byte global loads along C, then one thread per (column PAIR, 32-row block): 32 4-byte
// LDS reads down the pair, two amax / shared exponents, v_cvt_pk_fp8_f32, two 32-byte stores.  Rows R..Rpad-1 quantise as
// zeros (scale byte 0).  (Round 2 read one column per thread with 2-byte LDS reads from a 64-column tile: 3.5 TB/s.)
constexpr int TQ_COLS = 128, TQ_LD = 130;
__device__ __forceinline__ void tq_load_tile(bf16 (*tile)[TQ_LD], const bf16* __restrict__ x, int ldx, int r0, int c0, int R, int C, int t) {
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int r = (t >> 4) + 16 * p, cc = (t & 15) * 8;
    bf16x8 v{};
    if (r0 + r < R) {
      if (c0 + cc + 8 <= C) v = *reinterpret_cast<const bf16x8*>(x + (int64_t)(r0 + r) * ldx + c0 + cc);
      else for (int j = 0; j < 8; ++j) if (c0 + cc + j < C) v[j] = x[(int64_t)(r0 + r) * ldx + c0 + cc + j];
    }
#pragma unroll
    for (int j = 0; j < 8; j += 2) *reinterpret_cast<bf16x2*>(&tile[r][cc + j]) = bf16x2{v[j], v[j + 1]};
  }
}
// Transposed blocks of one thread: columns c0 + 2cp, +1; source rows r0 + 32 blk ..: quantised into registers (w[h][8] = 32
// bytes of output row c0 + 2cp + h, scale bytes sc[h]).
__device__ __forceinline__ void tq_quantize_pair(const bf16 (*tile)[TQ_LD], int t, uint32_t (&w)[2][8], uint8_t (&sc)[2]) {
  const int cp = t & 63, blk = t >> 6;
  float v0[32], v1[32];
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const bf16x2 e = *reinterpret_cast<const bf16x2*>(&tile[32 * blk + k][2 * cp]);
    v0[k] = (float)e[0]; v1[k] = (float)e[1];
    a0 = fmaxf(a0, fabsf(v0[k])); a1 = fmaxf(a1, fabsf(v1[k]));
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float (&v)[32] = h ? v1 : v0;
    const int E = mx_shared_exp(h ? a1 : a0);
    sc[h] = (uint8_t)(E + 127);
    const float inv = exp2f((float)-E);
#pragma unroll
    for (int i = 0; i < 8; ++i) w[h][i] = mx_pack4(v[i * 4], v[i * 4 + 1], v[i * 4 + 2], v[i * 4 + 3], inv);
  }
}
// The quantised tile leaves through LDS (the bf16 image is dead by then - callers synchronise first): out[c][128 source rows] as
// 144-byte rows plus the 4 scale bytes of every column, so that the global stores are whole 128-byte output-row segments (8 lanes
// x 16 bytes) instead of one 32-byte piece per lane scattered over 64 output rows.
constexpr int TQ_OUT_LD = 144;
__device__ __forceinline__ void tq_stage_and_store(char* stage, const uint32_t (&w)[2][8], const uint8_t (&sc)[2], uint8_t* __restrict__ q, int ldq,
                                                   uint8_t* __restrict__ s, int lds_, int r0, int c0, int C, int Rpad, int t) {
  const int cp = t & 63, blk = t >> 6;
  uint8_t* sstage = reinterpret_cast<uint8_t*>(stage) + TQ_COLS * TQ_OUT_LD;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    u32x4* o = reinterpret_cast<u32x4*>(stage + (2 * cp + h) * TQ_OUT_LD + 32 * blk);
    o[0] = u32x4{w[h][0], w[h][1], w[h][2], w[h][3]};
    o[1] = u32x4{w[h][4], w[h][5], w[h][6], w[h][7]};
    sstage[(2 * cp + h) * 4 + blk] = sc[h];
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int c = it * 32 + (t >> 3), seg = t & 7;
    if (c0 + c < C && r0 + 16 * seg < Rpad)
      *reinterpret_cast<u32x4*>(q + (int64_t)(c0 + c) * ldq + r0 + 16 * seg) = *reinterpret_cast<const u32x4*>(stage + c * TQ_OUT_LD + 16 * seg);
  }
  if (t < TQ_COLS && c0 + t < C) {
    uint8_t* sp = s + (int64_t)(c0 + t) * lds_ + (r0 >> 5);
    const int nb = min(4, (Rpad - r0) >> 5);
    if (nb == 4 && (lds_ & 3) == 0 && ((uintptr_t)s & 3) == 0) *reinterpret_cast<uint32_t*>(sp) = *reinterpret_cast<const uint32_t*>(sstage + t * 4);
    else for (int i = 0; i < nb; ++i) sp[i] = sstage[t * 4 + i];
  }
}
__global__ __launch_bounds__(256) void transpose_quantize_mxfp8_kernel(const bf16* __restrict__ x, int ldx, uint8_t* __restrict__ q, int ldq,
                                                                      uint8_t* __restrict__ s, int lds_, int R, int C, int Rpad) {
  __shared__ __attribute__((aligned(16))) bf16 tile[128][TQ_LD];
  static_assert(sizeof(bf16) * 128 * TQ_LD >= TQ_COLS * TQ_OUT_LD + TQ_COLS * 4, "the output staging reuses the tile's LDS");
  const int r0 = blockIdx.y * 128, c0 = blockIdx.x * TQ_COLS;
  const int t = threadIdx.x;
  tq_load_tile(tile, x, ldx, r0, c0, R, C, t);
  __syncthreads();
  uint32_t w[2][8]; uint8_t sc[2];
  tq_quantize_pair(tile, t, w, sc);
  __syncthreads();                                         // every thread is done reading the bf16 image
  tq_stage_and_store(reinterpret_cast<char*>(&tile[0][0]), w, sc, q, ldq, s, lds_, r0, c0, C, Rpad, t);
}
// ---------------------------------------------------------------- row-wise AND transposed quantisation in one pass
// What the backward of an fp8 linear needs of its dy (and the refresh of a weight needs of W): the row-wise quantisation
// (dgrad: dy . W^T) and the transposed one (wgrad: dy^T . x) - the same 128 x 128 tile through LDS, read from memory once
// (2 + 1 + 1 bytes per element instead of 2 + 1 and 2 + 1).  Every thread quantises its transposed column pair like
// transpose_quantize_mxfp8_kernel and two row blocks (row t >> 1, columns 64 (t & 1) + {0, 32} ..) like
// quantize_mxfp8_kernel: both outputs are bit-identical to the two separate kernels.  C % 64 == 0.
__global__ __launch_bounds__(256) void quantize_dual_mxfp8_kernel(const bf16* __restrict__ x, int ldx, uint8_t* __restrict__ q, int ldq,
                                                                 uint8_t* __restrict__ s, int lds_, uint8_t* __restrict__ qt, int ldqt,
                                                                 uint8_t* __restrict__ st, int ldst, int R, int C, int Rpad) {
  __shared__ __attribute__((aligned(16))) bf16 tile[128][TQ_LD];
  const int r0 = blockIdx.y * 128, c0 = blockIdx.x * TQ_COLS;
  const int t = threadIdx.x;
  tq_load_tile(tile, x, ldx, r0, c0, R, C, t);
  __syncthreads();
  uint32_t w[2][8]; uint8_t sc[2];
  tq_quantize_pair(tile, t, w, sc);
  const int r = t >> 1;
  if (r0 + r < R) {
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      const int cb = 64 * (t & 1) + 32 * hb;               // first column of this block inside the tile
      if (c0 + cb >= C) break;
      float v[32];
      float amax = 0.f;
#pragma unroll
      for (int k = 0; k < 32; k += 2) {
        const bf16x2 e = *reinterpret_cast<const bf16x2*>(&tile[r][cb + k]);
        v[k] = (float)e[0]; v[k + 1] = (float)e[1];
        amax = fmaxf(amax, fmaxf(fabsf(v[k]), fabsf(v[k + 1])));
      }
      const int E = mx_shared_exp(amax);
      s[(int64_t)(r0 + r) * lds_ + ((c0 + cb) >> 5)] = (uint8_t)(E + 127);
      const float inv = exp2f((float)-E);
      uint32_t wr[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) wr[i] = mx_pack4(v[i * 4], v[i * 4 + 1], v[i * 4 + 2], v[i * 4 + 3], inv);
      u32x4* qp = reinterpret_cast<u32x4*>(q + (int64_t)(r0 + r) * ldq + c0 + cb);
      qp[0] = u32x4{wr[0], wr[1], wr[2], wr[3]};
      qp[1] = u32x4{wr[4], wr[5], wr[6], wr[7]};
    }
  }
  __syncthreads();                                         // every thread is done reading the bf16 image
  tq_stage_and_store(reinterpret_cast<char*>(&tile[0][0]), w, sc, qt, ldqt, st, ldst, r0, c0, C, Rpad, t);
}
}  // namespace

#ifdef VLB_TOOLS
extern "C" void vlb_gemm_mxfp8_set_variant(int v) { g_fp8_variant = v & 0xff; g_fp8_rows = v >> 8; }     // rows in bits 8.. (0: planner)
// tools build only: ONE v_mfma_scale_f32_16x16x128_f8f6f4 on caller-given per-lane registers (layout experiments)
namespace {
template <int OA, int OB>
__global__ void mfma_scale_probe_kernel(const int* a, const int* b, const int* sa, const int* sb, float* d) {
  const int l = threadIdx.x;
  i32x8 av, bv;
  for (int i = 0; i < 8; ++i) { av[i] = a[l * 8 + i]; bv[i] = b[l * 8 + i]; }
  f32x4 c = {d[l * 4], d[l * 4 + 1], d[l * 4 + 2], d[l * 4 + 3]};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, OA, sa[l], OB, sb[l]);
  for (int i = 0; i < 4; ++i) d[l * 4 + i] = c[i];
}
}  // namespace
extern "C" int vlb_mfma_scale_probe(const int* a, const int* b, const int* sa, const int* sb, float* d, int mode, void* stream) {
  hipStream_t st = as_stream(stream);
  if (mode == 0) hipLaunchKernelGGL((mfma_scale_probe_kernel<0, 0>), dim3(1), dim3(64), 0, st, a, b, sa, sb, d);
  if (mode == 1) hipLaunchKernelGGL((mfma_scale_probe_kernel<0, 1>), dim3(1), dim3(64), 0, st, a, b, sa, sb, d);
  if (mode == 2) hipLaunchKernelGGL((mfma_scale_probe_kernel<1, 0>), dim3(1), dim3(64), 0, st, a, b, sa, sb, d);
  if (mode == 3) hipLaunchKernelGGL((mfma_scale_probe_kernel<2, 3>), dim3(1), dim3(64), 0, st, a, b, sa, sb, d);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
#endif

extern "C" int vlb_quantize_mxfp8(const void* x_bf16, int ldx, void* q, int ldq, void* scales, int lds, int rows, int K, void* stream) {
  VLB_REQUIRE(x_bf16 && q && scales && rows > 0 && K > 0 && K % 32 == 0 && ldx % 8 == 0 && ldq % 16 == 0 && ldx >= K && ldq >= K && lds >= K / 32,
              "quantize_mxfp8: K must be a multiple of 32, ldx of 8, ldq of 16");
  VLB_REQUIRE((((uintptr_t)x_bf16 | (uintptr_t)q) % 16) == 0, "quantize_mxfp8: 16-byte alignment required");
  const int kb = K / 32;
  const int64_t total = (int64_t)rows * kb;
  hipLaunchKernelGGL(quantize_mxfp8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), (const bf16*)x_bf16, ldx,
                     (uint8_t*)q, ldq, (uint8_t*)scales, lds, rows, kb);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

extern "C" int vlb_transpose_quantize_mxfp8(const void* x_bf16, int ldx, void* q, int ldq, void* scales, int lds, int R, int C, int Rpad,
                                            void* stream) {
  VLB_REQUIRE(x_bf16 && q && scales && R > 0 && C > 0 && Rpad >= R && Rpad % 32 == 0 && ldx % 8 == 0 && ldx >= C && ldq % 16 == 0 &&
                  ldq >= Rpad && lds >= Rpad / 32, "transpose_quantize_mxfp8: Rpad must be a multiple of 32, ldx of 8, ldq of 16");
  VLB_REQUIRE((((uintptr_t)x_bf16 | (uintptr_t)q) % 16) == 0, "transpose_quantize_mxfp8: 16-byte alignment required");
  dim3 grid((C + TQ_COLS - 1) / TQ_COLS, (Rpad + 127) / 128);
  hipLaunchKernelGGL(transpose_quantize_mxfp8_kernel, grid, dim3(256), 0, as_stream(stream), (const bf16*)x_bf16, ldx, (uint8_t*)q, ldq,
                     (uint8_t*)scales, lds, R, C, Rpad);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

extern "C" int vlb_quantize_dual_mxfp8(const void* x_bf16, int ldx, void* q, int ldq, void* scales, int lds, void* qt, int ldqt,
                                       void* scales_t, int ldst, int R, int C, int Rpad, void* stream) {
  VLB_REQUIRE(x_bf16 && q && scales && qt && scales_t && R > 0 && C > 0 && C % 64 == 0 && Rpad >= R && Rpad % 32 == 0,
              "quantize_dual_mxfp8: C must be a multiple of 64, Rpad of 32 (R=%d C=%d Rpad=%d)", R, C, Rpad);
  VLB_REQUIRE(ldx % 8 == 0 && ldx >= C && ldq % 16 == 0 && ldq >= C && lds >= C / 32 && ldqt % 16 == 0 && ldqt >= Rpad && ldst >= Rpad / 32,
              "quantize_dual_mxfp8: bad leading dimensions");
  VLB_REQUIRE((((uintptr_t)x_bf16 | (uintptr_t)q | (uintptr_t)qt) % 16) == 0, "quantize_dual_mxfp8: 16-byte alignment required");
  dim3 grid((C + TQ_COLS - 1) / TQ_COLS, (Rpad + 127) / 128);
  hipLaunchKernelGGL(quantize_dual_mxfp8_kernel, grid, dim3(256), 0, as_stream(stream), (const bf16*)x_bf16, ldx, (uint8_t*)q, ldq,
                     (uint8_t*)scales, lds, (uint8_t*)qt, ldqt, (uint8_t*)scales_t, ldst, R, C, Rpad);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

extern "C" int vlb_gemm_mxfp8(const void* Aq, int lda, const void* sA, int ldsa, const void* Wq, int ldw, const void* sW, int ldsw, void* C,
                              int ldc, int M, int N, int K, const void* residual, int ldr, void* stream) {
  VLB_REQUIRE(Aq && sA && Wq && sW && C, "gemm_mxfp8: null operand");
  VLB_REQUIRE(M > 0 && N > 0 && N % 256 == 0 && K > 0 && K % 128 == 0, "gemm_mxfp8: needs N %% 256 == 0 and K %% 128 == 0 (M=%d N=%d K=%d)", M, N, K);
  VLB_REQUIRE(lda % 16 == 0 && ldw % 16 == 0 && lda >= K && ldw >= K && ldsa >= K / 32 && ldsw >= K / 32 && ldsa % 4 == 0 && ldsw % 4 == 0 &&
                  ldc % 4 == 0 && ldc >= N, "gemm_mxfp8: bad leading dimensions (scale rows must be 4-byte multiples)");
  VLB_REQUIRE((((uintptr_t)Aq | (uintptr_t)Wq) % 16) == 0 && ((uintptr_t)C % 8) == 0 && (((uintptr_t)sA | (uintptr_t)sW) % 4) == 0,
              "gemm_mxfp8: misaligned operand");
  if (residual) VLB_REQUIRE(ldr >= N && ldr % 4 == 0 && ((uintptr_t)residual % 8) == 0, "gemm_mxfp8: bad residual");
  VLB_REQUIRE((int64_t)M * lda < (1ll << 32) && (int64_t)N * ldw < (1ll << 32), "gemm_mxfp8: operands above 4 GiB are not supported (32-bit staging offsets)");
  // One workgroup per CU: the GEMM runs in rounds of 256 tiles.  A partial last round that is at most 5/8 full is re-cut into
  // 256 x 128 (or 192 x 128) halves in a second launch (twice the workgroups, about 0.62 of a round).  192-row tiles are picked
  // when they quantise the row count into cheaper rounds: cost = rounds x tile rows, the smaller tile charged 2 % more (the rule
  // of the bf16 kernel, gemm.hip plan_rows, without split-K tails).
  hipStream_t st = as_stream(stream);
  const int cus = 256, tn = N / 256;
  auto plan = [&](int rows, int& tiles, int& rem, bool& halves) {
    tiles = ((M + rows - 1) / rows) * tn; rem = tiles % cus;
    halves = tiles > cus / 2 && rem > 0 && rem <= cus * 5 / 8;
    return (tiles / cus + (rem == 0 ? 0.0 : halves ? 0.62 : 1.0)) * rows;
  };
  int t256, r256, t192, r192; bool h256, h192;
  const double c256 = plan(256, t256, r256, h256), c192 = plan(192, t192, r192, h192) * 1.02;
  bool use192 = t256 > cus && c192 < 0.97 * c256;
#ifdef VLB_TOOLS
  if (g_fp8_rows == 256) use192 = false;
  if (g_fp8_rows == 192) use192 = true;
  if (g_fp8_variant != 1) use192 = false;
#endif
  const int rows = use192 ? 192 : 256, tiles = use192 ? t192 : t256, rem = use192 ? r192 : r256;
  const bool halves = use192 ? h192 : h256;
  Fp8Args a{(const uint8_t*)Aq, (const uint8_t*)sA, (const uint8_t*)Wq, (const uint8_t*)sW, (bf16*)C, (const bf16*)residual,
            M, N, K, lda, ldw, ldc, ldr, ldsa, ldsw, (M + rows - 1) / rows, tn};
  if (halves) {
    if (tiles - rem > 0) {
      int rc = use192 ? launch_fp8<8, 6>(a, Fp8Launch{0, 1}, tiles - rem, st) : launch_fp8<8>(a, Fp8Launch{0, 1}, tiles - rem, st);
      if (rc != VLB_OK) return rc;
    }
    return use192 ? launch_fp8<4, 6>(a, Fp8Launch{tiles - rem, 2}, 2 * rem, st) : launch_fp8<4>(a, Fp8Launch{tiles - rem, 2}, 2 * rem, st);
  }
  return use192 ? launch_fp8<8, 6>(a, Fp8Launch{0, 1}, tiles, st) : launch_fp8<8>(a, Fp8Launch{0, 1}, tiles, st);
}
