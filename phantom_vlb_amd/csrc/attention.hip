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

struct AttnArgs {
  const bf16* q; const bf16* k; const bf16* v; bf16* o; float* lse;
  const uint8_t* mask;
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
  const int qb = CAUSAL ? (nqb - 1 - blockIdx.x) : blockIdx.x;   // heavy (late) causal blocks first
  const int hq = blockIdx.y, b = blockIdx.z;
  const int hkv = hq / (p.Hq / p.Hkv);
  const int q0 = qb * QB + wave * QW;    // first query row of this wave
  const int ql = lane & 31, h = lane >> 5;

  // ---- Q fragments (B operand): lane holds Q[q0+ql][16ks + 8h + j]
  bf16x8 qf[KS];
  {
    const int qr = min(q0 + ql, p.S - 1);
    const bf16* qp = p.q + ((int64_t)b * p.S + qr) * p.ldq + hq * D + 8 * h;
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
  const int q_hi = min(qb * QB + QB, p.S) - 1;
  const int ntiles = CAUSAL ? (q_hi / KV + 1) : (p.S + KV - 1) / KV;

  const bf16* kbase = p.k + (int64_t)b * p.S * p.ldk + hkv * D;
  const bf16* vbase = p.v + (int64_t)b * p.S * p.ldv + hkv * D;

  // ---- K/V staging by LDS-DMA: wave-instruction i of wave w fills LDS bytes [(4i+w)*1024, +1024) of
  // the tile; the swizzle is applied on the per-lane SOURCE chunk, the LDS image stays lane-linear.
  constexpr int RPI = 1024 / ROWB;       // rows per wave-instruction (4 for D=128, 8 for D=64)
  const int sr = lane / CPR, sp = lane % CPR;
  auto stage = [&](int buf, int t) {
    char* kb = smem + buf * 2 * TILE; char* vb = kb + TILE;
#pragma unroll
    for (int i = 0; i < LD; ++i) {
      const int r = (4 * i + wave) * RPI + sr;
      const int key = min(t * KV + r, p.S - 1);   // clamp: rows past the end are masked, V stays finite
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
      // ---- masks: key validity (padding + sequence end) as one 64-bit word, causal on diagonal tiles
      unsigned long long kvalid;
      {
        const int key = key0 + lane;
        bool ok = key < p.S;
        if (ok && p.mask) ok = p.mask[(int64_t)b * p.S + key] != 0;
        kvalid = __ballot(ok);
      }
      const bool diag = CAUSAL && (key0 + KV - 1 > q0);
      const bool partial = diag || (kvalid != ~0ull);
      const int qrow = q0 + ql;
      float mloc = NEG;
      if (partial) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kl = 32 * s + (r & 3) + 8 * (r >> 2) + 4 * h;
            bool ok = (kvalid >> kl) & 1ull;
            if (CAUSAL) ok = ok && (key0 + kl <= qrow);
            st[s][r] = ok ? st[s][r] : NEG;
          }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[s][r]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float m_new = fmaxf(m_run, mloc);
      const float alpha = exp2f((m_run - m_new) * c2);
      const float mc = m_new * c2;
      float lsum = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float e = exp2f(st[s][r] * c2 - mc);
          if (partial) e = (st[s][r] <= NEG) ? 0.f : e;
          st[s][r] = e;
          lsum += e;
        }
      l_run = l_run * alpha + lsum;
      m_run = m_new;
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;
      // ---- O^T += V^T . P^T ; P^T fragment of k-step (s, s2) = bf16(st[s][8*s2 .. 8*s2+7])
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
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            const bf16x8 vf = __builtin_bit_cast(bf16x8, both);
            ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, ot[dt], 0, 0, 0);
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
  if (qrow < p.S) {
    bf16* op = p.o + ((int64_t)b * p.S + qrow) * p.ldo + hq * D;
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

template <int D, bool CAUSAL>
int launch_fwd(const AttnArgs& a, hipStream_t s) {
  constexpr int LDS = 2 * 2 * KV * D * 2;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<D, CAUSAL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) { vlb_set_error("attention: LDS reservation failed: %s", hipGetErrorString(e)); return VLB_ERR_LAUNCH; }
    configured = true;
  }
  dim3 grid((a.S + QB - 1) / QB, a.Hq, a.B);
  hipLaunchKernelGGL((attn_fwd_kernel<D, CAUSAL>), grid, dim3(256), LDS, s, a);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}
}  // namespace

extern "C" int vlb_attention_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out,
                                 int ldo, float* lse, const uint8_t* key_mask, int B, int S, int Hq, int Hkv, int D,
                                 int causal, float scale, void* stream) {
  VLB_REQUIRE(q && k && v && out, "attention_fwd: null operand");
  VLB_REQUIRE(D == 64 || D == 128, "attention_fwd: head dim %d not in {64,128}", D);
  VLB_REQUIRE(B > 0 && S > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "attention_fwd: bad shape B=%d S=%d Hq=%d Hkv=%d", B, S, Hq, Hkv);
  VLB_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "attention_fwd: strides must keep 16-byte alignment");
  VLB_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 == 0) && ((uintptr_t)out % 8 == 0), "attention_fwd: misaligned pointer");
  AttnArgs a{(const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)out, lse, key_mask, ldq, ldk, ldv, ldo, B, S, Hq, Hkv, scale};
  hipStream_t s = as_stream(stream);
  if (D == 128) return causal ? launch_fwd<128, true>(a, s) : launch_fwd<128, false>(a, s);
  return causal ? launch_fwd<64, true>(a, s) : launch_fwd<64, false>(a, s);
}
