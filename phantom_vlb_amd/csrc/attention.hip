// Flash-style attention forward/backward for gfx950 (bf16 in/out, fp32 softmax state).
//
// Forward structure: a workgroup = 4 waves = 128 query rows of one (batch, q-head); each wave owns
// 32 rows.  K/V tiles of 64 keys are register-staged into double-buffered, XOR-swizzled LDS images.
// Scores are computed TRANSPOSED (S^T = K.Q^T with v_mfma_f32_32x32x16_bf16) so a query row lives
// on one lane pair (l, l+32): row max / row sum need a single cross-half exchange, and the S^T
// accumulator registers are, after bf16 packing, directly the B operand of O^T += V^T.P^T.
// V^T fragments come from the row-major V image with ds_read_b64_tr_b16 (hardware transpose).
#include "common.hpp"

namespace {

constexpr int KV = 64;     // keys per tile
constexpr int QW = 32;     // query rows per wave
constexpr int NW = 4;      // waves per workgroup
constexpr int QB = QW * NW;
constexpr float NEG = -1e30f;

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                   (void __attribute__((address_space(3)))*)l, 16, 0, 0);
}

typedef short s16x4 __attribute__((ext_vector_type(4)));

// ds_read_b64_tr_b16 through inline asm.  hipcc treats the builtin as possibly aliasing a pending LDS-DMA
// and drains vmcnt(0) in front of it (which here also waits for the fp32 dQ atomics); the asm form is
// invisible to that bookkeeping, so every batch of reads below is followed by an explicit lgkmcnt(0)
// and a sched_barrier before its first consumer (cdna_hip_programming.md 5.7, form iii).
__device__ __forceinline__ s16x4 tr_read_asm(const char* lds_ptr) {
  s16x4 v;
  const unsigned a = (unsigned)(uintptr_t)(const char __attribute__((address_space(3)))*)lds_ptr;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(a));
  return v;
}
__device__ __forceinline__ void lds_wait_all() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ bf16x8 join8(s16x4 lo, s16x4 hi) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

struct AttnArgs {
  const bf16* q; const bf16* k; const bf16* v; bf16* o; float* lse;
  const uint8_t* mask;
  const int* cu;           // packed rows: clip b = rows [cu[b], cu[b+1]); NULL = dense b*S
  int ldq, ldk, ldv, ldo;
  int B, S, Hq, Hkv;
  float scale;
};

template <int D> __device__ __forceinline__ int slot_k(int r, int c) {
  if constexpr (D == 128) return c ^ (r & 15); else return c ^ ((r >> 1) & 7);
}
template <int D> __device__ __forceinline__ int slot_v(int r, int c) {
  if constexpr (D == 128) return c ^ ((r & 3) << 2); else return c ^ (((r >> 1) & 1) << 2);
}

// Head index of the j-th workgroup of a (block, clip) segment.  Consecutive workgroups land on consecutive XCDs
// (id mod 8), each with its own L2: with Hq a multiple of 8, XCD x gets q-heads x*Hq/8 .. - i.e. the q-heads of a
// GQA group (which stream the same K/V tiles at the same time) share one L2 instead of fetching them 4 times.
__device__ __forceinline__ int xcd_head(int j, int Hq) {
  return (Hq & 7) == 0 ? (j & 7) * (Hq >> 3) + (j >> 3) : j;
}

template <int D, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs p) {
  constexpr int CPR = D / 8;             // 16-byte chunks per row
  constexpr int ROWB = D * 2;            // bytes per row
  constexpr int TILE = KV * ROWB;        // bytes per K or V tile
  constexpr int LD = (KV * CPR) / 256;   // 16-byte chunks each thread stages per tile
  constexpr int KS = D / 16;             // k-steps of QK^T
  constexpr int DT = D / 32;             // d-tiles of O^T
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K tile | V tile]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = (p.S + QB - 1) / QB;
  // 1-D grid ordered by weight over the whole launch: with a causal mask every workgroup of the last query block
  // (the most key tiles) is dispatched before any of the previous one, for all heads and clips
  const int per_qb = p.Hq * p.B, slot = blockIdx.x / per_qb;
  const int qb = CAUSAL ? (nqb - 1 - slot) : slot;
  const int hq = xcd_head((blockIdx.x % per_qb) % p.Hq, p.Hq), b = (blockIdx.x % per_qb) / p.Hq;
  const int hkv = hq / (p.Hq / p.Hkv);
  const int Sb = p.cu ? p.cu[b + 1] - p.cu[b] : p.S;                 // tokens of this clip
  const int64_t row0 = p.cu ? p.cu[b] : (int64_t)b * p.S;            // its first row
  if (qb * QB >= Sb) return;                                          // whole workgroup past the clip's end
  const int q0 = qb * QB + wave * QW;    // first query row of this wave
  const int ql = lane & 31, h = lane >> 5;

  // ---- Q fragments (B operand): lane holds Q[q0+ql][16ks + 8h + j]
  bf16x8 qf[KS];
  {
    const int qr = min(q0 + ql, Sb - 1);
    const bf16* qp = p.q + (row0 + qr) * p.ldq + hq * D + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
  }

  f32x16 ot[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[i][r] = 0.f;
  float m_run = NEG, l_run = 0.f;
  const float c2 = p.scale * 1.44269504088896341f;

  // number of KV tiles this workgroup needs
  const int q_hi = min(qb * QB + QB, Sb) - 1;
  const int ntiles = CAUSAL ? (q_hi / KV + 1) : (Sb + KV - 1) / KV;

  const bf16* kbase = p.k + row0 * p.ldk + hkv * D;
  const bf16* vbase = p.v + row0 * p.ldv + hkv * D;

  // ---- K/V staging by LDS-DMA: wave-instruction i of wave w fills LDS bytes [(4i+w)*1024, +1024) of
  // the tile; the swizzle is applied on the per-lane SOURCE chunk, the LDS image stays lane-linear.
  constexpr int RPI = 1024 / ROWB;       // rows per wave-instruction (4 for D=128, 8 for D=64)
  const int sr = lane / CPR, sp = lane % CPR;
  auto stage = [&](int buf, int t) {
    char* kb = smem + buf * 2 * TILE; char* vb = kb + TILE;
#pragma unroll
    for (int i = 0; i < LD; ++i) {
      const int r = (4 * i + wave) * RPI + sr;
      const int key = min(t * KV + r, Sb - 1);   // clamp: rows past the end are masked, V stays finite
      glds16(kbase + (int64_t)key * p.ldk + slot_k<D>(r, sp) * 8, kb + (4 * i + wave) * 1024);
      glds16(vbase + (int64_t)key * p.ldv + slot_v<D>(r, sp) * 8, vb + (4 * i + wave) * 1024);
    }
  };

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // per-lane LDS read bases: K row reads (one per k-step, subtile via +32 rows) and V transposed reads
  // (one per d-tile; key rows via compile-time offsets)
  const int g1 = (lane >> 4) & 1, li = lane & 15, tq = li >> 2, tp = li & 3;
  int k_rd[KS], v_rd[DT];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) k_rd[ks] = ql * ROWB + slot_k<D>(ql, 2 * ks + h) * 16;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    const int r0 = 4 * h + tq;           // + 32s + 16s2 (+8): multiples of 8 never change the V swizzle
    v_rd[dt] = r0 * ROWB + slot_v<D>(r0, 4 * dt + 2 * g1 + (tp >> 1)) * 16 + (tp & 1) * 8;
  }

  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    const bool more = (t + 1 < ntiles);
    if (more) stage(cur ^ 1, t + 1);
    const char* kb = smem + cur * 2 * TILE; const char* vb = kb + TILE;
    const int key0 = t * KV;
    // a wave whose rows all precede this tile has nothing to do here (causal)
    const bool active = !CAUSAL || (key0 <= q0 + QW - 1);
    if (active) {
      // ---- S^T = K . Q^T
      f32x16 st[2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[s][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb + k_rd[ks] + 32 * s * ROWB);
          st[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st[s], 0, 0, 0);
        }
      }
      // ---- masks: key validity (padding + sequence end) as one 64-bit word, causal on diagonal tiles.
      // The masked path is a separate wave-uniform branch (most tiles are full and skip it entirely).
      unsigned long long kvalid;
      {
        const int key = key0 + lane;
        bool ok = key < Sb;
        if (ok && p.mask) ok = p.mask[row0 + key] != 0;
        kvalid = __ballot(ok);
      }
      const bool diag = CAUSAL && (key0 + KV - 1 > q0);
      const bool partial = diag || (kvalid != ~0ull);
      if (partial) {
        const int qrow = q0 + ql;
        const unsigned lo = (unsigned)(kvalid >> (4 * h)), hi = (unsigned)(kvalid >> (32 + 4 * h));
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const unsigned word = s ? hi : lo;           // bit (r&3) + 8(r>>2) of `word` = key 32s + 4h + ...
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kb = (r & 3) + 8 * (r >> 2);
            bool ok = (word >> kb) & 1u;
            if (CAUSAL) ok = ok && (key0 + 32 * s + 4 * h + kb <= qrow);
            st[s][r] = ok ? st[s][r] : NEG;
          }
        }
        asm volatile("" ::: "memory");                 // keep this block a real branch (no if-conversion)
      }
      float mloc = fmaxf(st[0][0], st[1][0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) mloc = fmaxf(fmaxf(mloc, st[0][r]), st[1][r]);   // v_max3_f32
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float m_new = fmaxf(m_run, mloc);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
      const float mc = m_new * c2;
      float lsum = 0.f;
      // exp2(NEG*c2 - mc) underflows to exactly 0 unless the whole row is still masked (m_new == NEG):
      // then every entry would read exp2(0) = 1, so zero the row explicitly in that (rare) case.
      const float live = (m_new > 0.5f * NEG) ? 1.f : 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(st[s][r] * c2 - mc);
          st[s][r] = e;
          lsum += e;
        }
      if (partial && live == 0.f) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int r = 0; r < 16; ++r) st[s][r] = 0.f;
        lsum = 0.f;
      }
      l_run = l_run * alpha + lsum;
      m_run = m_new;
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;
      // ---- O^T += V^T . P^T ; P^T fragment of k-step (s, s2) = bf16(st[s][8*s2 .. 8*s2+7]).
      // (builtin transposed reads: the compiler pipelines them against the MFMAs; the asm form used in
      // the backward measured 10 % slower here because every batch needs a full lgkmcnt(0))
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          bf16x8 pf;
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[j] = (bf16)st[s][8 * s2 + j];
          const int roff = (32 * s + 16 * s2) * ROWB;   // key rows for elements j=0..3 ; +8 rows for j=4..7
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(vb + v_rd[dt] + roff));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(vb + v_rd[dt] + roff + 8 * ROWB));
            ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join8(lo, hi), pf, ot[dt], 0, 0, 0);
          }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
  const int qrow = q0 + ql;
  if (qrow < Sb) {
    bf16* op = p.o + (row0 + qrow) * p.ldo + hq * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)(ot[dt][4 * g + e] * inv);
        *reinterpret_cast<bf16x4*>(op + 32 * dt + 8 * g + 4 * h) = o;
      }
    if (p.lse && h == 0)
      p.lse[((int64_t)b * p.Hq + hq) * p.S + qrow] = l_tot > 0.f ? m_run * p.scale + logf(l_tot) : -INFINITY;
  }
}

#ifdef VLB_TOOLS
// -------------------------------------------------------------------------------------------------
// Forward, 64 query rows per wave (round 4 EXPERIMENT, tools build only: measured 39-45 % SLOWER, see below).
// The kernel above is LDS-bound: per 64-key tile a wave reads 16 K fragments
// (ds_read_b128) and 32 transposed V^T fragments (ds_read_b64_tr_b16) = 32 KB for 32 MFMAs, and with eight waves on a CU
// that is ~250 B/clk against the 256 B/clk the LDS delivers.  Here a wave owns TWO 32-row query blocks and every K and
// V^T fragment it reads feeds both (half the LDS bytes per MFMA); a workgroup = 4 waves = 256 query rows, ONE per CU, one
// wave per SIMD with the 512-register budget (2 x 64 accumulator registers of O^T, 2 x 32 of S^T, 2 x 32 registers of Q
// fragments).  Same arithmetic per row as the kernel above (same tile order, same online-softmax updates): bit-identical.
// Measured (tools/bench_attention.py, interleaved): decoder B=3 231.8 vs 140.3 us, B=5 356.4 vs 218.3, ViT D=64 165.5 vs 90.9 -
// with ONE compiler-scheduled wave per SIMD nothing covers a fragment read's latency or the softmax's VALU work (the two
// waves per SIMD of the kernel above cover each other's); halving the LDS bytes does not pay for that.  The structure needs
// a hand-placed instruction stream (cdna guide, "4-wave, one-wave-per-SIMD" forward), which was not attempted.
// -------------------------------------------------------------------------------------------------
template <int D, bool CAUSAL>
__global__ __launch_bounds__(256, 1) void attn_fwd64_kernel(AttnArgs p) {
  constexpr int CPR = D / 8, ROWB = D * 2, TILE = KV * ROWB, LD = (KV * CPR) / 256, KS = D / 16, DT = D / 32;
  constexpr int NQ = 2, QWW = 32 * NQ, QBB = QWW * NW;          // 64 rows per wave, 256 per workgroup
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K tile | V tile]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = (p.S + QBB - 1) / QBB;
  const int per_qb = p.Hq * p.B, slot = blockIdx.x / per_qb;
  const int qb = CAUSAL ? (nqb - 1 - slot) : slot;
  const int hq = xcd_head((blockIdx.x % per_qb) % p.Hq, p.Hq), b = (blockIdx.x % per_qb) / p.Hq;
  const int hkv = hq / (p.Hq / p.Hkv);
  const int Sb = p.cu ? p.cu[b + 1] - p.cu[b] : p.S;
  const int64_t row0 = p.cu ? p.cu[b] : (int64_t)b * p.S;
  if (qb * QBB >= Sb) return;
  const int q0 = qb * QBB + wave * QWW;  // first query row of this wave (a multiple of 64: both blocks see the same tiles)
  const int ql = lane & 31, h = lane >> 5;

  bf16x8 qf[NQ][KS];
#pragma unroll
  for (int u = 0; u < NQ; ++u) {
    const int qr = min(q0 + 32 * u + ql, Sb - 1);
    const bf16* qp = p.q + (row0 + qr) * p.ldq + hq * D + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[u][ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
  }
  f32x16 ot[NQ][DT];
#pragma unroll
  for (int u = 0; u < NQ; ++u)
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[u][i][r] = 0.f;
  float m_run[NQ], l_run[NQ];
#pragma unroll
  for (int u = 0; u < NQ; ++u) { m_run[u] = NEG; l_run[u] = 0.f; }
  const float c2 = p.scale * 1.44269504088896341f;

  const int q_hi = min(qb * QBB + QBB, Sb) - 1;
  const int ntiles = CAUSAL ? (q_hi / KV + 1) : (Sb + KV - 1) / KV;
  const bf16* kbase = p.k + row0 * p.ldk + hkv * D;
  const bf16* vbase = p.v + row0 * p.ldv + hkv * D;
  constexpr int RPI = 1024 / ROWB;
  const int sr = lane / CPR, sp = lane % CPR;
  auto stage = [&](int buf, int t) {
    char* kb = smem + buf * 2 * TILE; char* vb = kb + TILE;
#pragma unroll
    for (int i = 0; i < LD; ++i) {
      const int r = (4 * i + wave) * RPI + sr;
      const int key = min(t * KV + r, Sb - 1);
      glds16(kbase + (int64_t)key * p.ldk + slot_k<D>(r, sp) * 8, kb + (4 * i + wave) * 1024);
      glds16(vbase + (int64_t)key * p.ldv + slot_v<D>(r, sp) * 8, vb + (4 * i + wave) * 1024);
    }
  };
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int g1 = (lane >> 4) & 1, li = lane & 15, tq = li >> 2, tp = li & 3;
  int k_rd[KS], v_rd[DT];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) k_rd[ks] = ql * ROWB + slot_k<D>(ql, 2 * ks + h) * 16;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    const int r0 = 4 * h + tq;
    v_rd[dt] = r0 * ROWB + slot_v<D>(r0, 4 * dt + 2 * g1 + (tp >> 1)) * 16 + (tp & 1) * 8;
  }

  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntiles) stage(cur ^ 1, t + 1);
    const char* kb = smem + cur * 2 * TILE; const char* vb = kb + TILE;
    const int key0 = t * KV;
    const bool active = !CAUSAL || (key0 <= q0 + QWW - 1);
    if (active) {
      // ---- S^T = K . Q^T for both query blocks: one K fragment, two MFMAs
      f32x16 st[NQ][2];
#pragma unroll
      for (int u = 0; u < NQ; ++u)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int r = 0; r < 16; ++r) st[u][s][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb + k_rd[ks] + 32 * s * ROWB);
#pragma unroll
          for (int u = 0; u < NQ; ++u) st[u][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[u][ks], st[u][s], 0, 0, 0);
        }
      }
      unsigned long long kvalid;
      {
        const int key = key0 + lane;
        bool ok = key < Sb;
        if (ok && p.mask) ok = p.mask[row0 + key] != 0;
        kvalid = __ballot(ok);
      }
      float alpha[NQ];
#pragma unroll
      for (int u = 0; u < NQ; ++u) {
        const int qu0 = q0 + 32 * u;
        const bool diag = CAUSAL && (key0 + KV - 1 > qu0);
        const bool partial = diag || (kvalid != ~0ull);
        if (partial) {
          const int qrow = qu0 + ql;
          const unsigned lo = (unsigned)(kvalid >> (4 * h)), hi = (unsigned)(kvalid >> (32 + 4 * h));
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const unsigned word = s ? hi : lo;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int kb2 = (r & 3) + 8 * (r >> 2);
              bool ok = (word >> kb2) & 1u;
              if (CAUSAL) ok = ok && (key0 + 32 * s + 4 * h + kb2 <= qrow);
              st[u][s][r] = ok ? st[u][s][r] : NEG;
            }
          }
          asm volatile("" ::: "memory");
        }
        float mloc = fmaxf(st[u][0][0], st[u][1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(fmaxf(mloc, st[u][0][r]), st[u][1][r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run[u], mloc);
        alpha[u] = __builtin_amdgcn_exp2f((m_run[u] - m_new) * c2);
        const float mc = m_new * c2;
        float lsum = 0.f;
        const float live = (m_new > 0.5f * NEG) ? 1.f : 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(st[u][s][r] * c2 - mc);
            st[u][s][r] = e;
            lsum += e;
          }
        if (partial && live == 0.f) {
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[u][s][r] = 0.f;
          lsum = 0.f;
        }
        l_run[u] = l_run[u] * alpha[u] + lsum;
        m_run[u] = m_new;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) ot[u][i][r] *= alpha[u];
      }
      // ---- O^T += V^T . P^T: one transposed V^T fragment, two MFMAs
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          bf16x8 pf[NQ];
#pragma unroll
          for (int u = 0; u < NQ; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[u][j] = (bf16)st[u][s][8 * s2 + j];
          const int roff = (32 * s + 16 * s2) * ROWB;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(vb + v_rd[dt] + roff));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(vb + v_rd[dt] + roff + 8 * ROWB));
            const bf16x8 vt = join8(lo, hi);
#pragma unroll
            for (int u = 0; u < NQ; ++u) ot[u][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt, pf[u], ot[u][dt], 0, 0, 0);
          }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

#pragma unroll
  for (int u = 0; u < NQ; ++u) {
    const float l_tot = l_run[u] + __shfl_xor(l_run[u], 32, 64);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    const int qrow = q0 + 32 * u + ql;
    if (qrow < Sb) {
      bf16* op = p.o + (row0 + qrow) * p.ldo + hq * D;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16)(ot[u][dt][4 * g + e] * inv);
          *reinterpret_cast<bf16x4*>(op + 32 * dt + 8 * g + 4 * h) = o;
        }
      if (p.lse && h == 0)
        p.lse[((int64_t)b * p.Hq + hq) * p.S + qrow] = l_tot > 0.f ? m_run[u] * p.scale + logf(l_tot) : -INFINITY;
    }
  }
}

int g_attn_fwd64 = 0;          // tools: bit 0 = 64-row waves in the forward for D = 128 (decoder), bit 1 = for D = 64 (ViT)

template <int D, bool CAUSAL>
int launch_fwd64(const AttnArgs& a, hipStream_t s) {
  constexpr int LDS = 2 * 2 * KV * D * 2;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd64_kernel<D, CAUSAL>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  if (attr != hipSuccess) { vlb_set_error("attention: LDS reservation failed: %s", hipGetErrorString(attr)); return VLB_ERR_LAUNCH; }
  dim3 grid(((a.S + 255) / 256) * a.Hq * a.B);
  hipLaunchKernelGGL((attn_fwd64_kernel<D, CAUSAL>), grid, dim3(256), LDS, s, a);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

#endif  // VLB_TOOLS (64-rows-per-wave forward experiment)

template <int D, bool CAUSAL>
int launch_fwd(const AttnArgs& a, hipStream_t s) {
  constexpr int LDS = 2 * 2 * KV * D * 2;
  // once per process and kernel; a function-local static's initialisation is thread-safe (C++11)
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<D, CAUSAL>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  if (attr != hipSuccess) { vlb_set_error("attention: LDS reservation failed: %s", hipGetErrorString(attr)); return VLB_ERR_LAUNCH; }
  dim3 grid(((a.S + QB - 1) / QB) * a.Hq * a.B);
  hipLaunchKernelGGL((attn_fwd_kernel<D, CAUSAL>), grid, dim3(256), LDS, s, a);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
}  // namespace

extern "C" int vlb_attention_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out,
                                 int ldo, float* lse, const uint8_t* key_mask, int B, int S, int Hq, int Hkv, int D,
                                 int causal, float scale, const int* cu_rows, void* stream) {
  VLB_REQUIRE(q && k && v && out, "attention_fwd: null operand");
  VLB_REQUIRE(D == 64 || D == 128, "attention_fwd: head dim %d not in {64,128}", D);
  VLB_REQUIRE(B > 0 && S > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "attention_fwd: bad shape B=%d S=%d Hq=%d Hkv=%d", B, S, Hq, Hkv);
  VLB_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "attention_fwd: strides must keep 16-byte alignment");
  VLB_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 == 0) && ((uintptr_t)out % 8 == 0), "attention_fwd: misaligned pointer");
  AttnArgs a{(const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)out, lse, key_mask, cu_rows, ldq, ldk, ldv, ldo, B, S, Hq, Hkv, scale};
  hipStream_t s = as_stream(stream);
#ifdef VLB_TOOLS
  if ((int64_t)((S + 255) / 256) * Hq * B >= 256) {          // A/B of the 64-rows-per-wave experiment
    if (D == 128 && (g_attn_fwd64 & 1)) return causal ? launch_fwd64<128, true>(a, s) : launch_fwd64<128, false>(a, s);
    if (D == 64 && (g_attn_fwd64 & 2)) return causal ? launch_fwd64<64, true>(a, s) : launch_fwd64<64, false>(a, s);
  }
#endif
  if (D == 128) return causal ? launch_fwd<128, true>(a, s) : launch_fwd<128, false>(a, s);
  return causal ? launch_fwd<64, true>(a, s) : launch_fwd<64, false>(a, s);
}
#ifdef VLB_TOOLS
extern "C" void vlb_attn_set_fwd64(int on) { g_attn_fwd64 = on; }
#endif

// =================================================================================================
// Backward (D = 128 only: the decoder; the frozen ViT never needs one).
//
// A workgroup = 4 waves owns 128 keys of one (batch, kv-head) and sweeps the q-heads of its GQA
// group x 32-row query blocks, so dK/dV accumulate in registers with no cross-workgroup sum.
// Scores are computed with the KEY ON THE LANE (S = Q.K^T, dP = dO.V^T): their accumulators are then
// directly the B operands of dV^T += dO^T.P and dK^T += Q^T.dS (transposed A fragments come from the
// row-major Q/dO LDS images via ds_read_b64_tr_b16).  Only dS crosses LDS (image [key][q]) so that
// each wave can form one 32-column slice of dQ over all 128 keys; dQ is summed across key blocks
// with fp32 atomics (two 128-byte segments per wave-instruction), then converted to bf16.
// =================================================================================================
namespace {

constexpr int BK_KEYS = 128;   // keys per workgroup
constexpr int BQ = 32;         // queries per inner block

// The product library runs ONE backward: per-q-head dK/dV workgroups + GQA reduce + dQ pass.  The superseded 8-wave
// per-kv-head kernel (with or without fp32 dQ atomics) and its timing-only ablations exist only in the tools build
// (libvlb_tools.so, -DVLB_TOOLS) for tools/bench_attention.py's A/B.
#ifdef VLB_TOOLS
int g_attn_ablate = 0;
int g_attn_split_dq = 1;      // 1 (default): per-q-head dK/dV workgroups + dQ pass; 2: the 8-wave per-kv-head dK/dV kernel + dQ pass;
                              // 0: 8-wave kernel with fp32 dQ atomics (A/B only)
#else
constexpr int g_attn_ablate = 0, g_attn_split_dq = 1;
#endif
struct AttnBwdArgs {
  const bf16* q; const bf16* k; const bf16* v; const bf16* dout;
  const float* lse; const float* delta; const uint8_t* mask; const int* cu;
  bf16* dk; bf16* dv; float* dq_acc;
  int ldq, ldk, ldv, lddo, lddk, lddv;
  int B, S, Hq, Hkv;
  float scale;
  int ablate;      // timing-only experiments (wrong results): bit0 skip dQ atomics, bit1 skip the whole dQ phase
  int split_dq;    // dQ is computed by attn_bwd_dq_kernel: this kernel only produces dK / dV
};

__device__ __forceinline__ int sw2(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

// delta[b,h,s] = sum_d dout*out ; one 16-lane group per (row, head).  Packed rows are mapped back to
// (clip, position) through cu.
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16* __restrict__ o, int ldo, const bf16* __restrict__ d_o,
                                                         int lddo, float* __restrict__ delta, int S, int Hq, int B,
                                                         const int* __restrict__ cu, int64_t total) {
  const int64_t gid = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);   // (row, head) index
  const int c = threadIdx.x & 15;
  float s = 0.f;
  if (gid < total) {
    const int h = gid % Hq; const int64_t tok = gid / Hq;
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(o + tok * ldo + h * 128 + c * 8);
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(d_o + tok * lddo + h * 128 + c * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (float)a[i] * (float)b[i];
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (gid < total && c == 0) {
    const int h = gid % Hq; const int64_t tok = gid / Hq;
    int64_t b = tok / S; int sidx = tok % S;
    if (cu) {
      int bb = 0;
      while (bb + 1 < B && tok >= cu[bb + 1]) ++bb;
      b = bb; sidx = (int)(tok - cu[bb]);
    }
    delta[(b * Hq + h) * S + sidx] = s;
  }
}

#ifdef VLB_TOOLS
// 8 waves per workgroup: waves 0-3 and 4-7 ("groups") own the SAME 4 x 32 keys but sweep different
// q-heads of the GQA group, so every SIMD holds two waves whose MFMA / LDS / VALU / atomic phases
// overlap; K and V tiles are shared, Q/dO/dS^T/lse buffers are per group, and the two partial
// dK/dV accumulators are summed through LDS once at the end.
template <bool CAUSAL>
__global__ __launch_bounds__(512, 2) void attn_bwd_kernel(AttnBwdArgs p) {
  constexpr int D = 128, ROWB = 256;
  constexpr int K_OFF = 0, V_OFF = BK_KEYS * ROWB;                 // 32 KB each, shared by both groups
  constexpr int QT_BYTES = BQ * ROWB;                              // 8 KB
  constexpr int G_BASE = 2 * BK_KEYS * ROWB;                       // per-group region starts here
  constexpr int G_QT = 0;                                          // [2][Q tile | dO tile] = 32 KB
  constexpr int G_T = 4 * QT_BYTES;                                // dS^T image [128 keys][32 q] bf16 = 8 KB (chunk-swizzled)
  constexpr int G_L = G_T + BK_KEYS * 64;                          // [2][lse 32 | delta 32] f32 = 512 B
  constexpr int G_BYTES = G_L + 512;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3;
  char* gsm = smem + G_BASE + grp * G_BYTES;
  const int kb = blockIdx.x, hkv = blockIdx.y, b = blockIdx.z;
  const int k0 = kb * BK_KEYS;
  const int Sb = p.cu ? p.cu[b + 1] - p.cu[b] : p.S;
  const int64_t row0 = p.cu ? p.cu[b] : (int64_t)b * p.S;
  if (k0 >= Sb) return;                           // no keys of this clip in the block (uniform: before any barrier)
  const int gsz = p.Hq / p.Hkv;
  const int h_lo = grp == 0 ? 0 : (gsz + 1) / 2;                   // this group's q-heads inside the GQA group
  const int h_n = grp == 0 ? (gsz + 1) / 2 : gsz / 2;
  const int ql = lane & 31, h = lane >> 5;
  const int g1 = (lane >> 4) & 1, li = lane & 15, tq = li >> 2, tp = li & 3;
  const int sr = lane >> 4, sp = lane & 15;

  const bf16* kbase = p.k + row0 * p.ldk + hkv * D;
  const bf16* vbase = p.v + row0 * p.ldv + hkv * D;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = 8 * i + wave, r = piece * 4 + sr;
    const int key = min(k0 + r, Sb - 1);
    glds16(kbase + (int64_t)key * p.ldk + (sp ^ sw2(r)) * 8, smem + K_OFF + piece * 1024);
    glds16(vbase + (int64_t)key * p.ldv + (sp ^ sw2(r)) * 8, smem + V_OFF + piece * 1024);
  }
  const int nqb = (Sb + BQ - 1) / BQ;
  const int qb0 = CAUSAL ? (k0 / BQ) : 0;
  const int nq = nqb - qb0;
  const int my_total = nq * h_n;                                   // items of this group
  const int total = nq * ((gsz + 1) / 2);                          // loop count (group 0 has >= group 1)

  auto stage_q = [&](int buf, int item) {
    const int hq = hkv * gsz + h_lo + item / nq, qb = qb0 + item % nq;
    const bf16* qbase = p.q + row0 * p.ldq + hq * D;
    const bf16* dobase = p.dout + row0 * p.lddo + hq * D;
    char* qt = gsm + G_QT + buf * 2 * QT_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = 4 * i + w4, r = piece * 4 + sr;
      const int qr = min(qb * BQ + r, Sb - 1);
      glds16(qbase + (int64_t)qr * p.ldq + (sp ^ sw2(r)) * 8, qt + piece * 1024);
      glds16(dobase + (int64_t)qr * p.lddo + (sp ^ sw2(r)) * 8, qt + QT_BYTES + piece * 1024);
    }
    if (w4 == 0) {
      const int qr = min(qb * BQ + (lane & 31), Sb - 1);
      const float* src = (lane < 32 ? p.lse : p.delta) + ((int64_t)b * p.Hq + hq) * p.S + qr;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                       (void __attribute__((address_space(3)))*)(gsm + G_L + buf * 256), 4, 0, 0);
    }
  };

  f32x16 dkt[4], dvt[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkt[i][r] = 0.f; dvt[i][r] = 0.f; }

  if (my_total > 0) stage_q(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int krow = w4 * 32 + ql;
  int kv_rd[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) kv_rd[ks] = krow * ROWB + (((2 * ks + h) ^ sw2(krow)) << 4);
  int q_rd[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) q_rd[ks] = ql * ROWB + (((2 * ks + h) ^ sw2(ql)) << 4);
  int qt_rd[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int r = 8 * e + 4 * h + tq;
      qt_rd[dt][e] = r * ROWB + (((4 * dt + 2 * g1 + (tp >> 1)) ^ sw2(r)) << 4) + (tp & 1) * 8;
    }
  int kt_rd[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int r = 8 * h + 4 * e + tq;
    kt_rd[e] = r * ROWB + (((4 * w4 + 2 * g1 + (tp >> 1)) ^ sw2(r)) << 4) + (tp & 1) * 8;
  }
  // dS^T image rows are 64 bytes (8 chunks of 8 B); chunk c of row R sits at c ^ ((R>>2)&7) so that neither the
  // 8-byte writes (32 consecutive rows, one chunk) nor the transposed reads (4 rows x 8 chunks) collide on banks.
  // Row R = 16s + 8h + 4e + tq  ->  (R>>2)&7 = 4(s&1) + 2h + e: two offsets per e, picked by the parity of s.
  int t_rd[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int sp1 = 0; sp1 < 2; ++sp1)
      t_rd[e][sp1] = (8 * h + 4 * e + tq) * 64 + (((4 * g1 + tp) ^ (4 * sp1 + 2 * h + e)) << 3);

  const float c2 = p.scale * 1.44269504088896341f;
  const int key = k0 + krow;
  bool key_ok = key < Sb;
  if (key_ok && p.mask) key_ok = p.mask[row0 + key] != 0;

  for (int item = 0; item < total; ++item) {
    const int cur = item & 1;
    const bool active = item < my_total;            // group-uniform; inactive waves only keep the barriers
    if (item + 1 < my_total) stage_q(cur ^ 1, item + 1);
    const int hq = hkv * gsz + h_lo + item / nq, qb = qb0 + item % nq;
    const char* qt = gsm + G_QT + cur * 2 * QT_BYTES;
    const char* dot = qt + QT_BYTES;
    const float* ls = reinterpret_cast<const float*>(gsm + G_L) + cur * 64;
    const int q0 = qb * BQ;
    bf16x8 pb[2], dsb[2];
    if (active) {
      // ---- S = Q.K^T and dP = dO.V^T  (key on lane)
      f32x16 sacc, pacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(qt + q_rd[ks]);
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(smem + K_OFF + kv_rd[ks]);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, kf, sacc, 0, 0, 0);
        const bf16x8 dof = *reinterpret_cast<const bf16x8*>(dot + q_rd[ks]);
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(smem + V_OFF + kv_rd[ks]);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof, vf, pacc, 0, 0, 0);
      }
      // ---- P = exp(S*scale - lse), dS = scale * P * (dP - delta); rows q = (r&3) + 8(r>>2) + 4h
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(ls + 8 * g + 4 * h);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(ls + 32 + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const int qrow = q0 + 8 * g + 4 * h + e;
          bool ok = key_ok && qrow < Sb;
          if (CAUSAL) ok = ok && (key <= qrow);
          const float pv = ok ? __builtin_amdgcn_exp2f(sacc[r] * c2 - l4[e] * 1.44269504088896341f) : 0.f;
          const float ds = pv * (pacc[r] - d4[e]) * p.scale;
          pb[r >> 3][r & 7] = (bf16)pv;
          dsb[r >> 3][r & 7] = (bf16)ds;
        }
      }
      if (!p.split_dq) {
        // ---- dS^T image: row = key (krow), cols q = 8g + 4h + {0..3}  (8-byte stores)
  #pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 t;
  #pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = dsb[g >> 1][(g & 1) * 4 + e];
          *reinterpret_cast<bf16x4*>(gsm + G_T + krow * 64 + (((2 * g + h) ^ ((krow >> 2) & 7)) << 3)) = t;
        }
      }
      // ---- dV^T += dO^T.P ; dK^T += Q^T.dS   (contraction over the 32 queries, 2 k-steps)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        s16x4 ra[4][2], rb[4][2];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            ra[dt][e] = tr_read_asm(dot + qt_rd[dt][e] + 16 * s * ROWB);
            rb[dt][e] = tr_read_asm(qt + qt_rd[dt][e] + 16 * s * ROWB);
          }
        lds_wait_all();
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dvt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join8(ra[dt][0], ra[dt][1]), pb[s], dvt[dt], 0, 0, 0);
          dkt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join8(rb[dt][0], rb[dt][1]), dsb[s], dkt[dt], 0, 0, 0);
        }
      }
    }
    // raw barrier: __syncthreads() would add vmcnt(0) and drain the in-flight DMA and atomics
    if (!p.split_dq) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // dS^T images complete
    }
    if (p.split_dq) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // next Q/dO tile landed
    } else if (active && !(p.ablate & 2)) {
      // ---- dQ[:, 32*w4 .. +32] = dS . K  over the workgroup's 128 keys (8 k-steps of 16 keys)
      f32x16 dq;
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
      for (int sh = 0; sh < 2; ++sh) {
        s16x4 ta[4][2], tk[4][2];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const int s = 4 * sh + s4;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            ta[s4][e] = tr_read_asm(gsm + G_T + t_rd[e][s & 1] + 16 * s * 64);
            tk[s4][e] = tr_read_asm(smem + K_OFF + kt_rd[e] + 16 * s * ROWB);
          }
        }
        lds_wait_all();
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join8(ta[s4][0], ta[s4][1]), join8(tk[s4][0], tk[s4][1]), dq, 0, 0, 0);
      }
      // 16 no-return fp32 atomics per wave, ALWAYS issued (rows past the end add 0 to the last valid row)
      float* dqp = p.dq_acc + (row0 * p.Hq + hq) * D + 32 * w4 + ql;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = q0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const bool ok = qi < Sb;
        if (!(p.ablate & 1)) atomicAdd(dqp + (int64_t)min(qi, Sb - 1) * p.Hq * D, ok ? dq[r] : 0.f);
        else if (dq[r] == 12345.678f) dqp[0] = 1.f;      // keep the MFMAs alive
      }
      // the next tile's DMA (issued at the top of this iteration) is older than the 16 atomics:
      // vmcnt(16) retires it and leaves the atomics in flight across the barrier.
      if (!(p.ablate & 1)) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // next Q/dO tile landed; T and the current tile are free again
  }

  // ---- sum the two groups' dK^T/dV^T through LDS (all tiles are dead now), group 0 writes the result
  float* red = reinterpret_cast<float*>(smem);       // [w4][8 tiles][16 regs][64 lanes] fp32 = 128 KB
  if (grp == 1) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        red[((w4 * 8 + dt) * 16 + r) * 64 + lane] = dkt[dt][r];
        red[((w4 * 8 + 4 + dt) * 16 + r) * 64 + lane] = dvt[dt][r];
      }
  }
  __syncthreads();
  if (grp == 0 && key < Sb) {
    bf16* dkp = p.dk + (row0 + key) * p.lddk + hkv * D;
    bf16* dvp = p.dv + (row0 + key) * p.lddv + hkv * D;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 a, c;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          a[e] = (bf16)(dkt[dt][r] + red[((w4 * 8 + dt) * 16 + r) * 64 + lane]);
          c[e] = (bf16)(dvt[dt][r] + red[((w4 * 8 + 4 + dt) * 16 + r) * 64 + lane]);
        }
        *reinterpret_cast<bf16x4*>(dkp + 32 * dt + 8 * g + 4 * h) = a;
        *reinterpret_cast<bf16x4*>(dvp + 32 * dt + 8 * g + 4 * h) = c;
      }
  }
}

#endif  // VLB_TOOLS (superseded 8-wave dK/dV kernel)

// -------------------------------------------------------------------------------------------------
// dK / dV pass, one workgroup per (128 keys, q-head).  The causal triangle gives the first key block 16x the work
// of the last; sweeping all q-heads of a GQA group inside one workgroup (the 8-wave kernel above) makes that
// workgroup the critical path.  Here every q-head is its own workgroup (4x more, 4x shorter, heaviest first), the
// per-head partials are written as bf16 - which is also what autograd does for the reference's repeat_kv, whose
// backward sums the expanded heads in bf16 - and attn_dkdv_reduce_kernel adds them in a fixed order.
// 4 waves x 32 keys, K fragments in REGISTERS for the whole kernel, V tile in LDS once, 32-row Q/dO tiles through a
// double-buffered LDS-DMA ring; 65 KB of LDS: two workgroups per CU; one barrier per 32 queries.
// -------------------------------------------------------------------------------------------------
template <bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(AttnBwdArgs p, bf16* __restrict__ part) {
  constexpr int D = 128, ROWB = 256;
  constexpr int V_OFF = 0;                                          // V tile: 128 keys x 256 B = 32 KB
  constexpr int QT_BYTES = BQ * ROWB;                               // 8 KB
  constexpr int Q_OFF = BK_KEYS * ROWB;                             // [2][Q tile | dO tile] = 32 KB
  constexpr int L_OFF = Q_OFF + 4 * QT_BYTES;                       // [2][lse 32 | delta 32] f32 = 512 B
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 1-D grid ordered by weight: every workgroup of key block 0 (the most query blocks under the causal mask) is
  // dispatched before any of key block 1, and so on - longest jobs first over the whole launch
  const int per_kb = p.Hq * p.B;
  const int kb = blockIdx.x / per_kb, hq = xcd_head((blockIdx.x % per_kb) % p.Hq, p.Hq), b = (blockIdx.x % per_kb) / p.Hq;
  const int gsz = p.Hq / p.Hkv;
  const int hkv = hq / gsz;
  const int k0 = kb * BK_KEYS;
  const int Sb = p.cu ? p.cu[b + 1] - p.cu[b] : p.S;
  const int64_t row0 = p.cu ? p.cu[b] : (int64_t)b * p.S;
  if (k0 >= Sb) return;
  const int ql = lane & 31, h = lane >> 5;
  const int g1 = (lane >> 4) & 1, li = lane & 15, tq = li >> 2, tp = li & 3;
  const int sr = lane >> 4, sp = lane & 15;

  const int krow = w4 * 32 + ql;
  const int key = k0 + krow;
  // K fragments (B operand of S = Q.K^T): lane holds K[key][16ks + 8h + j]
  bf16x8 kf[8];
  {
    const bf16* kp = p.k + (row0 + min(key, Sb - 1)) * p.ldk + hkv * D + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(kp + 16 * ks);
  }
  const bf16* vbase = p.v + row0 * p.ldv + hkv * D;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int piece = 4 * i + w4, r = piece * 4 + sr;
    glds16(vbase + (int64_t)min(k0 + r, Sb - 1) * p.ldv + (sp ^ sw2(r)) * 8, smem + V_OFF + piece * 1024);
  }
  const int nqb = (Sb + BQ - 1) / BQ;
  const int qb0 = CAUSAL ? (k0 / BQ) : 0;
  const int total = nqb - qb0;
  const bf16* qbase = p.q + row0 * p.ldq + hq * D;
  const bf16* dobase = p.dout + row0 * p.lddo + hq * D;
  const float* lsebase = p.lse + ((int64_t)b * p.Hq + hq) * p.S;
  const float* delbase = p.delta + ((int64_t)b * p.Hq + hq) * p.S;

  auto stage_q = [&](int buf, int item) {
    const int qb = qb0 + item;
    char* qt = smem + Q_OFF + buf * 2 * QT_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = 4 * i + w4, r = piece * 4 + sr;
      const int qr = min(qb * BQ + r, Sb - 1);
      glds16(qbase + (int64_t)qr * p.ldq + (sp ^ sw2(r)) * 8, qt + piece * 1024);
      glds16(dobase + (int64_t)qr * p.lddo + (sp ^ sw2(r)) * 8, qt + QT_BYTES + piece * 1024);
    }
    if (w4 == 0) {
      const int qr = min(qb * BQ + (lane & 31), Sb - 1);
      const float* src = (lane < 32 ? lsebase : delbase) + qr;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                       (void __attribute__((address_space(3)))*)(smem + L_OFF + buf * 256), 4, 0, 0);
    }
  };

  f32x16 dkt[4], dvt[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkt[i][r] = 0.f; dvt[i][r] = 0.f; }

  if (total > 0) stage_q(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // row-read offsets are formed at the use site: chunk (2ks + h) ^ sw2(row) = (2ks) ^ (h ^ sw2(row)), 2ks even
  const int v_base = krow * ROWB, v_x = h ^ sw2(krow);
  const int q_base = ql * ROWB, q_x = h ^ sw2(ql);
  int qt_rd[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int r = 8 * e + 4 * h + tq;
      qt_rd[dt][e] = r * ROWB + (((4 * dt + 2 * g1 + (tp >> 1)) ^ sw2(r)) << 4) + (tp & 1) * 8;
    }

  const float c2 = p.scale * 1.44269504088896341f;
  bool key_ok = key < Sb;
  if (key_ok && p.mask) key_ok = p.mask[row0 + key] != 0;

  for (int item = 0; item < total; ++item) {
    const int cur = item & 1;
    if (item + 1 < total) stage_q(cur ^ 1, item + 1);
    const char* qt = smem + Q_OFF + cur * 2 * QT_BYTES;
    const char* dot = qt + QT_BYTES;
    const float* ls = reinterpret_cast<const float*>(smem + L_OFF) + cur * 64;
    const int q0 = (qb0 + item) * BQ;
    const bool active = !CAUSAL || (k0 + w4 * 32 <= q0 + BQ - 1);     // else: this wave's keys all follow the block
    if (active) {
      f32x16 sacc, pacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(qt + q_base + (((2 * ks) ^ q_x) << 4));
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, kf[ks], sacc, 0, 0, 0);
        const bf16x8 dof = *reinterpret_cast<const bf16x8*>(dot + q_base + (((2 * ks) ^ q_x) << 4));
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(smem + V_OFF + v_base + (((2 * ks) ^ v_x) << 4));
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof, vf, pacc, 0, 0, 0);
      }
      bf16x8 pb[2], dsb[2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(ls + 8 * g + 4 * h);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(ls + 32 + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const int qrow = q0 + 8 * g + 4 * h + e;
          bool ok = key_ok && qrow < Sb;
          if (CAUSAL) ok = ok && (key <= qrow);
          const float pv = ok ? __builtin_amdgcn_exp2f(sacc[r] * c2 - l4[e] * 1.44269504088896341f) : 0.f;
          const float ds = pv * (pacc[r] - d4[e]) * p.scale;
          pb[r >> 3][r & 7] = (bf16)pv;
          dsb[r >> 3][r & 7] = (bf16)ds;
        }
      }
      // dV^T += dO^T.P ; dK^T += Q^T.dS: the transposed fragments of one operand at a time (live set < 256 registers)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        s16x4 ra[4][2];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int e = 0; e < 2; ++e) ra[dt][e] = tr_read_asm(dot + qt_rd[dt][e] + 16 * s * ROWB);
        lds_wait_all();
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          dvt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join8(ra[dt][0], ra[dt][1]), pb[s], dvt[dt], 0, 0, 0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int e = 0; e < 2; ++e) ra[dt][e] = tr_read_asm(qt + qt_rd[dt][e] + 16 * s * ROWB);
        lds_wait_all();
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          dkt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join8(ra[dt][0], ra[dt][1]), dsb[s], dkt[dt], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // next Q/dO tile landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // ... for every wave, and the current tile is free again
  }

  if (key < Sb) {
    // a lone head per kv-head (no GQA) writes dK / dV directly, otherwise its bf16 partial [row][Hq][dK 128 | dV 128]
    bf16* dkp = gsz == 1 ? p.dk + (row0 + key) * p.lddk + hkv * D : part + ((row0 + key) * p.Hq + hq) * (2 * D);
    bf16* dvp = gsz == 1 ? p.dv + (row0 + key) * p.lddv + hkv * D : dkp + D;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 a, c;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = (bf16)dkt[dt][4 * g + e]; c[e] = (bf16)dvt[dt][4 * g + e]; }
        *reinterpret_cast<bf16x4*>(dkp + 32 * dt + 8 * g + 4 * h) = a;
        *reinterpret_cast<bf16x4*>(dvp + 32 * dt + 8 * g + 4 * h) = c;
      }
  }
}

// dk / dv [row][hkv*128 + d] = sum over the GQA group's q-heads of their bf16 partials (fp32 sum, fixed order)
__global__ __launch_bounds__(256) void attn_dkdv_reduce_kernel(const bf16* __restrict__ part, bf16* __restrict__ dk, int lddk,
                                                               bf16* __restrict__ dv, int lddv, int Hq, int Hkv, int64_t total) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;          // (row, hkv, which, 16-byte chunk of 128 d)
  if (i >= total) return;
  const int c = i & 15, which = (i >> 4) & 1;
  const int hkv = (int)((i >> 5) % Hkv);
  const int64_t row = (i >> 5) / Hkv;
  const int gsz = Hq / Hkv;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bf16* src = part + (row * Hq + (int64_t)hkv * gsz) * 256 + which * 128 + c * 8;
  for (int j = 0; j < gsz; ++j) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (int64_t)j * 256);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
  }
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16)acc[e];
  bf16* dst = which ? dv + row * lddv : dk + row * lddk;
  *reinterpret_cast<bf16x8*>(dst + hkv * 128 + c * 8) = o;
}

// -------------------------------------------------------------------------------------------------
// dQ in its own pass, shaped like the forward kernel: a workgroup owns 128 query rows of one q-head and
// sweeps the key tiles, so dQ accumulates in registers - no fp32 atomics, no scratch buffer, and a fixed
// summation order (bit-reproducible).  Per 64-key tile a wave recomputes S^T = K.Q^T and dP^T = V.dO^T
// (query on the lane, as in the forward), forms dS^T = P^T.(dP^T - delta).scale in registers, and uses
// those accumulators directly as the B operand of dQ^T += K^T.dS^T; K^T fragments are transposed reads of
// the same row-major K tile (dual-use swizzle sw2, also conflict-free for the row reads).
// -------------------------------------------------------------------------------------------------
struct AttnDqArgs {
  const bf16* q; const bf16* k; const bf16* v; const bf16* dout;
  const float* lse; const float* delta; const uint8_t* mask; const int* cu;
  bf16* dq;
  int ldq, ldk, ldv, lddo, lddq;
  int B, S, Hq, Hkv;
  float scale;
};

template <bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnDqArgs p) {
  constexpr int D = 128, ROWB = 256, TILE = KV * ROWB, KS = 8, DT = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K tile | V tile]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = (p.S + QB - 1) / QB;
  // 1-D grid ordered by weight over the whole launch: with a causal mask every workgroup of the last query block
  // (the most key tiles) is dispatched before any of the previous one, for all heads and clips
  const int per_qb = p.Hq * p.B, slot = blockIdx.x / per_qb;
  const int qb = CAUSAL ? (nqb - 1 - slot) : slot;
  const int hq = xcd_head((blockIdx.x % per_qb) % p.Hq, p.Hq), b = (blockIdx.x % per_qb) / p.Hq;
  const int hkv = hq / (p.Hq / p.Hkv);
  const int Sb = p.cu ? p.cu[b + 1] - p.cu[b] : p.S;
  const int64_t row0 = p.cu ? p.cu[b] : (int64_t)b * p.S;
  if (qb * QB >= Sb) return;
  const int q0 = qb * QB + wave * QW;
  const int ql = lane & 31, h = lane >> 5;
  const int qrow = q0 + ql;
  const int qr = min(qrow, Sb - 1);

  // Q and dO fragments (B operands): lane holds X[q][16ks + 8h + j]
  bf16x8 qf[KS], dof[KS];
  {
    const bf16* qp = p.q + (row0 + qr) * p.ldq + hq * D + 8 * h;
    const bf16* dp = p.dout + (row0 + qr) * p.lddo + hq * D + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
      dof[ks] = *reinterpret_cast<const bf16x8*>(dp + 16 * ks);
    }
  }
  const float lse_q = p.lse[((int64_t)b * p.Hq + hq) * p.S + qr];
  const float delta_q = p.delta[((int64_t)b * p.Hq + hq) * p.S + qr];
  const float c2 = p.scale * 1.44269504088896341f;
  // a fully masked row has lse = -inf: make every probability underflow to 0 instead of inf
  const float lse2 = (lse_q > -1e37f) ? lse_q * 1.44269504088896341f : 1e37f;

  f32x16 dqt[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqt[i][r] = 0.f;

  const int q_hi = min(qb * QB + QB, Sb) - 1;
  const int ntiles = CAUSAL ? (q_hi / KV + 1) : (Sb + KV - 1) / KV;
  const bf16* kbase = p.k + row0 * p.ldk + hkv * D;
  const bf16* vbase = p.v + row0 * p.ldv + hkv * D;

  const int sr = lane >> 4, sp = lane & 15;       // 4 rows x 16 chunks per wave-instruction
  auto stage = [&](int buf, int t) {
    char* kb = smem + buf * 2 * TILE; char* vb = kb + TILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (4 * i + wave) * 4 + sr;
      const int key = min(t * KV + r, Sb - 1);
      glds16(kbase + (int64_t)key * p.ldk + (sp ^ sw2(r)) * 8, kb + (4 * i + wave) * 1024);
      glds16(vbase + (int64_t)key * p.ldv + (sp ^ sw2(r)) * 8, vb + (4 * i + wave) * 1024);
    }
  };
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int g1 = (lane >> 4) & 1, li = lane & 15, tq = li >> 2, tp = li & 3;
  int row_rd[KS];                                   // row reads (A operand of the score MFMAs): key = 32s + ql
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) row_rd[ks] = ql * ROWB + (((2 * ks + h) ^ sw2(ql)) << 4);
  int tr_rd[DT][2];                                 // transposed reads of K: keys 4h + tq (+8), 32 columns of block dt
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int r = 4 * h + tq + 8 * e;             // + 32s + 16s2: multiples of 16 leave sw2 unchanged
      tr_rd[dt][e] = r * ROWB + (((4 * dt + 2 * g1 + (tp >> 1)) ^ sw2(r)) << 4) + (tp & 1) * 8;
    }

  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntiles) stage(cur ^ 1, t + 1);
    const char* kb = smem + cur * 2 * TILE; const char* vb = kb + TILE;
    const int key0 = t * KV;
    const bool active = !CAUSAL || (key0 <= q0 + QW - 1);
    if (active) {
      f32x16 st[2], dp[2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[s][r] = 0.f; dp[s][r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb + row_rd[ks] + 32 * s * ROWB);
          st[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st[s], 0, 0, 0);
          const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vb + row_rd[ks] + 32 * s * ROWB);
          dp[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], dp[s], 0, 0, 0);
        }
      }
      unsigned long long kvalid;
      {
        const int key = key0 + lane;
        bool ok = key < Sb;
        if (ok && p.mask) ok = p.mask[row0 + key] != 0;
        kvalid = __ballot(ok);
      }
      const bool diag = CAUSAL && (key0 + KV - 1 > q0);
      const bool partial = diag || (kvalid != ~0ull);
      if (partial) {
        const unsigned lo = (unsigned)(kvalid >> (4 * h)), hi = (unsigned)(kvalid >> (32 + 4 * h));
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const unsigned word = s ? hi : lo;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kb2 = (r & 3) + 8 * (r >> 2);
            bool ok = (word >> kb2) & 1u;
            if (CAUSAL) ok = ok && (key0 + 32 * s + 4 * h + kb2 <= qrow);
            st[s][r] = ok ? st[s][r] : NEG;
          }
        }
        asm volatile("" ::: "memory");
      }
      // dS^T = P^T (dP^T - delta) scale, P^T = exp2(S^T c2 - lse2); masked entries underflow to exactly 0
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(st[s][r] * c2 - lse2);
          st[s][r] = pv * (dp[s][r] - delta_q) * p.scale;
        }
      // dQ^T += K^T . dS^T ; dS^T fragment of k-step (s, s2) = bf16(st[s][8 s2 .. 8 s2 + 7])
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          bf16x8 df;
#pragma unroll
          for (int j = 0; j < 8; ++j) df[j] = (bf16)st[s][8 * s2 + j];
          const int roff = (32 * s + 16 * s2) * ROWB;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(kb + tr_rd[dt][0] + roff));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(kb + tr_rd[dt][1] + roff));
            dqt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join8(lo, hi), df, dqt[dt], 0, 0, 0);
          }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  if (qrow < Sb) {
    bf16* op = p.dq + (row0 + qrow) * p.lddq + hq * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)dqt[dt][4 * g + e];
        *reinterpret_cast<bf16x4*>(op + 32 * dt + 8 * g + 4 * h) = o;
      }
  }
}

#ifdef VLB_TOOLS
// dq (bf16, strided) = dq_acc (fp32 [B*S, Hq*128])
__global__ void dq_convert_kernel(const float* __restrict__ acc, bf16* __restrict__ dq, int lddq, int width, int64_t total) {
  const int cpr = width >> 3;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx / cpr; const int c = (idx % cpr) * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(acc + row * width + c);
    const f32x4 b2 = *reinterpret_cast<const f32x4*>(acc + row * width + c + 4);
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = (bf16)a[i]; o[4 + i] = (bf16)b2[i]; }
    *reinterpret_cast<bf16x8*>(dq + row * lddq + c) = o;
  }
}
#endif
}  // namespace

extern "C" int vlb_attention_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* out,
                                 int ldo, const void* dout, int lddo, const float* lse, const uint8_t* key_mask, void* dq,
                                 int lddq, void* dk, int lddk, void* dv, int lddv, float* delta, float* dq_acc, int B,
                                 int S, int Hq, int Hkv, int D, int causal, float scale, const int* cu_rows, int total_rows,
                                 void* stream) {
  VLB_REQUIRE(q && k && v && out && dout && lse && dq && dk && dv && delta, "attention_bwd: null operand");
  VLB_REQUIRE(dq_acc || g_attn_split_dq == 2 || (g_attn_split_dq == 1 && Hq == Hkv), "attention_bwd: the [rows, Hq, D] fp32-sized workspace is required");
  VLB_REQUIRE(D == 128, "attention_bwd: head dim %d unsupported (only 128: the decoder)", D);
  VLB_REQUIRE(B > 0 && S > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "attention_bwd: bad shape");
  VLB_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && lddo % 8 == 0 && lddq % 8 == 0 &&
                  lddk % 4 == 0 && lddv % 4 == 0, "attention_bwd: strides must keep vector alignment");
  hipStream_t st = as_stream(stream);
  const int64_t rows = cu_rows ? (int64_t)total_rows : (int64_t)B * S;     // packed: sum of clip lengths
  VLB_REQUIRE(rows > 0 && rows <= (int64_t)B * S, "attention_bwd: total_rows=%d out of range", total_rows);
  const int64_t th = rows * Hq;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((th + 15) / 16)), dim3(256), 0, st, (const bf16*)out, ldo,
                     (const bf16*)dout, lddo, delta, S, Hq, B, cu_rows, th);
  VLB_LAUNCH_CHECK();
  AttnBwdArgs a{(const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)dout, lse, delta, key_mask, cu_rows,
                (bf16*)dk, (bf16*)dv, dq_acc, ldq, ldk, ldv, lddo, lddk, lddv, B, S, Hq, Hkv, scale, g_attn_ablate, g_attn_split_dq};
  dim3 grid((S + BK_KEYS - 1) / BK_KEYS, Hkv, B);
#ifdef VLB_TOOLS
  if (g_attn_split_dq != 1) {        // superseded 8-wave per-kv-head kernel (A/B only)
    const int split = g_attn_split_dq;
    if (!split) {
      hipError_t e = hipMemsetAsync(dq_acc, 0, (size_t)th * D * sizeof(float), st);
      if (e != hipSuccess) { vlb_set_error("attention_bwd: memset failed: %s", hipGetErrorString(e)); return VLB_ERR_LAUNCH; }
    }
    constexpr int LDS = 2 * BK_KEYS * 256 + 2 * (4 * BQ * 256 + BK_KEYS * 64 + 512);   // 145 KB
    static const hipError_t a1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    static const hipError_t a2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (a1 != hipSuccess || a2 != hipSuccess) { vlb_set_error("attention_bwd: LDS reservation failed"); return VLB_ERR_LAUNCH; }
    if (causal) hipLaunchKernelGGL(attn_bwd_kernel<true>, grid, dim3(512), LDS, st, a);
    else hipLaunchKernelGGL(attn_bwd_kernel<false>, grid, dim3(512), LDS, st, a);
    VLB_LAUNCH_CHECK();
    if (!split) {
      const int64_t total = rows * (Hq * D / 8);
      int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(dq_convert_kernel, dim3(blocks), dim3(256), 0, st, dq_acc, (bf16*)dq, lddq, Hq * D, total);
      VLB_LAUNCH_CHECK();
      return VLB_OK;
    }
  } else
#endif
  {
    const int gsz = Hq / Hkv;
    VLB_REQUIRE(gsz == 1 || dq_acc, "attention_bwd: the workspace (fp32 [rows, Hq, D]) is required for grouped-query heads");
    VLB_REQUIRE(lddk % 8 == 0 && lddv % 8 == 0 && (((uintptr_t)dk | (uintptr_t)dv) % 16) == 0, "attention_bwd: dk/dv rows must be 16-byte aligned");
    constexpr int LDS_KV = BK_KEYS * 256 + 4 * BQ * 256 + 512;        // 65 KB: two workgroups per CU
    // once per process and kernel; a function-local static's initialisation is thread-safe (C++11)
    static const hipError_t k1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_KV);
    static const hipError_t k2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_KV);
    if (k1 != hipSuccess || k2 != hipSuccess) { vlb_set_error("attention_bwd: LDS reservation failed (dk/dv)"); return VLB_ERR_LAUNCH; }
    bf16* part = reinterpret_cast<bf16*>(dq_acc);        // rows*Hq*256 bf16 = the same bytes as fp32 [rows, Hq, 128]
    dim3 gridh(grid.x * Hq * B);                         // 1-D, heaviest key blocks first (see the kernel)
    if (causal) hipLaunchKernelGGL(attn_bwd_dkdv_kernel<true>, gridh, dim3(256), LDS_KV, st, a, part);
    else hipLaunchKernelGGL(attn_bwd_dkdv_kernel<false>, gridh, dim3(256), LDS_KV, st, a, part);
    VLB_LAUNCH_CHECK();
    if (gsz > 1) {
      const int64_t tot = rows * Hkv * 32;
      hipLaunchKernelGGL(attn_dkdv_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, part, (bf16*)dk, lddk,
                         (bf16*)dv, lddv, Hq, Hkv, tot);
      VLB_LAUNCH_CHECK();
    }
  }
  AttnDqArgs d{(const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)dout, lse, delta, key_mask, cu_rows,
               (bf16*)dq, ldq, ldk, ldv, lddo, lddq, B, S, Hq, Hkv, scale};
  constexpr int LDS_DQ = 2 * 2 * KV * 256;       // 64 KB: two workgroups per CU
  static const hipError_t q1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DQ);
  static const hipError_t q2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DQ);
  if (q1 != hipSuccess || q2 != hipSuccess) { vlb_set_error("attention_bwd: LDS reservation failed (dq)"); return VLB_ERR_LAUNCH; }
  dim3 gq(((S + QB - 1) / QB) * Hq * B);
  if (causal) hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, gq, dim3(256), LDS_DQ, st, d);
  else hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, gq, dim3(256), LDS_DQ, st, d);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

#ifdef VLB_TOOLS
// tuning hook (not part of the stable ABI): timing-only ablations of the attention backward kernel
extern "C" void vlb_attn_set_ablation(int bits) { g_attn_ablate = bits & 3; g_attn_split_dq = (bits & 4) ? 0 : ((bits & 8) ? 2 : 1); }   // bit2: atomic dQ; bit3: 8-wave dK/dV + dQ pass
#endif
