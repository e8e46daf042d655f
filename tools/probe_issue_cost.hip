// Issue cost of operand-fetch instructions beside MFMAs at ONE wave per SIMD (the four-wave GEMM's regime).
// Standalone measurement program (tools only; not part of libvlb.so):
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/probe_issue_cost tools/probe_issue_cost.hip && tools/build/probe_issue_cost
//
// Every workgroup = 256 threads = 4 waves (one per SIMD, 128 KB of LDS so only one workgroup fits a CU), each wave runs
// ITERS iterations of 128 independent v_mfma_f32_16x16x32_bf16 (64 accumulator tiles, two sweeps = one K-tile of the
// 256x256x64 GEMM tile: 2048 matrix cycles) with N fetch instructions of one KIND spread evenly among them, and a counted
// wait at the end of each iteration for the loads of the PREVIOUS iteration.  Sources walk a [rows][K = 4096] bf16 matrix
// 128 bytes per iteration like the GEMM's K loop (8 KB row stride), from a region sized to stay in the XCD's L2.
// Reported: cycles per iteration (s_memtime, median over workgroups) -> slope = cycles per fetch instruction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { K_NONE = 0, K_GLDS = 1, K_GLOAD_FRAG = 2, K_GLOAD_LINE = 3, K_BUFLDS = 4, K_BUFLOAD_FRAG = 5, K_DSREAD = 6 };

__device__ __forceinline__ void mfma_tied(f32x4& c, const i32x4& a, const i32x4& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void probe(const char* __restrict__ src, float* out, unsigned long long* cyc, int iters, int ld_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x4 acc[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  i32x4 fa = {lane * 3 + 1, lane * 5 + 2, 0x3f803f80, 0x3f003f00}, fb = {0x3f803f80, lane + 7, 0x3e803e80, 0x3f803f80};
  // per-lane source offsets.  "line" shape: 8 rows x 128 B (what an LDS-DMA piece of the GEMM fetches);
  // "frag" shape: 16 rows x 64 B (an MFMA B fragment fetched straight into registers).
  const char* base = src + (size_t)(blockIdx.x % 64) * 256 * ld_bytes;      // 64 distinct 256-row panels (2 MB each at K = 4096)
  const uint32_t off_line = (uint32_t)((wave * 8 + (lane >> 3)) * ld_bytes + (lane & 7) * 16);
  const uint32_t off_frag = (uint32_t)((wave * 16 + (lane & 15)) * ld_bytes + (lane >> 4) * 16);        // + (f & 3) * 64 rows: <= row 255
  // destinations of the register-destination fetches: "+v" operands, so each stays allocated to ITS register for the whole loop
  // (an "=v" destination is dead to the compiler until the next definition: it re-used the register for an address while the
  // load was still in flight, and the late write-back turned that address into garbage - a memory fault, not a timing)
  u32x4 sink[N > 0 ? N : 1];
#pragma unroll
  for (int i = 0; i < (N > 0 ? N : 1); ++i) sink[i] = u32x4{0, 0, 0, 0};
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 256 * ld_bytes, 0x00020000);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const uint32_t koff = (uint32_t)((it * 128) % 8192);            // walk the first 8 KB of every row whatever the stride
    // counted wait, as in the GEMM: the N fetches of the PREVIOUS iteration stay in flight, everything older is complete
    // (destinations are overwritten while in flight - harmless here, nothing reads them)
    if constexpr (KIND != K_NONE && KIND != K_DSREAD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N > 63 ? 63 : N) : "memory");
    if constexpr (KIND == K_DSREAD) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N > 15 ? 15 : N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 32; ++g) {
      // fetch instruction f is issued in front of group f * 32 / N
      if constexpr (N > 0) {
        if (g % (32 / (N > 32 ? 32 : N)) == 0) {
          constexpr int PER = N > 32 ? N / 32 : 1;
#pragma unroll
          for (int q = 0; q < PER; ++q) {
            const int f = (g / (32 / (N > 32 ? 32 : N))) * PER + q;
            const uint32_t row_blk = (uint32_t)((f & 7) * 32 * ld_bytes);           // line shape: 8 row blocks of 32 rows = 256 rows
            const uint32_t row_blk64 = (uint32_t)((f & 3) * 64 * ld_bytes);         // frag shape: 4 row blocks of 64 rows (16 per wave)
            if constexpr (KIND == K_GLDS) {
              __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + off_line + row_blk + koff),
                                               (void __attribute__((address_space(3)))*)(smem + (f & 31) * 4096 + wave * 1024), 16, 0, 0);
            } else if constexpr (KIND == K_BUFLDS) {
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (void __attribute__((address_space(3)))*)(smem + (f & 31) * 4096 + wave * 1024), 16,
                                                       off_line + row_blk, koff, 0, 0);
            } else if constexpr (KIND == K_GLOAD_FRAG) {
              asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(sink[f]) : "v"(base + off_frag + row_blk64 + koff + (f >> 2 & 1) * 64) : "memory");
            } else if constexpr (KIND == K_GLOAD_LINE) {
              asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(sink[f]) : "v"(base + off_line + row_blk + koff) : "memory");
            } else if constexpr (KIND == K_BUFLOAD_FRAG) {
              asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(sink[f]) : "v"(off_frag + row_blk64 + (f >> 2 & 1) * 64), "s"(rsrc), "s"(koff) : "memory");
            } else if constexpr (KIND == K_DSREAD) {
              asm volatile("ds_read_b128 %0, %1" : "+v"(sink[f]) : "v"((uint32_t)((f & 31) * 4096 + lane * 16)) : "memory");
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) mfma_tied(acc[(g * 4 + q) & 63], fa, fb);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (KIND != K_NONE) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) { asm volatile("" : "+a"(acc[i])); s += acc[i][0] + acc[i][3]; }
  unsigned x = 0;
#pragma unroll
  for (int i = 0; i < (N > 0 ? N : 1); ++i) { asm volatile("" : "+v"(sink[i])); x ^= sink[i][0] ^ sink[i][3]; }
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(x & 1) + (float)smem[threadIdx.x * 16];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// ---- the K loop of the four-wave GEMM as a skeleton: per iteration (= one 256x256x64 K-tile) two blocks of 64 MFMAs with one
// ds_read_b128 per 4 MFMAs (swizzled, conflict-free fragment reads of a 64 KB stage), the LDS-DMA of a later tile (16 pieces per
// wave, 8 rows x 128 B each, into the other stage), one barrier.  VAR selects what is left out / moved, to see which
// combination costs what when everything runs together (the single-kind rows above are each nearly free):
//   0 reads only             1 reads + barrier                 2 DMA only (block 2) + vmcnt(0) + barrier
//   3 = the GEMM: reads + DMA in block 2 + vmcnt(0) + barrier  4 like 3, DMA spread over both blocks (needs a 3rd stage in a real kernel)
//   5 like 3 without the barrier                               6 like 3 with a counted vmcnt(16) (tile t+1 waited one tile later)
//   7 like 3 with only 8 DMA pieces (the W-direct kernel's LDS side)
template <int VAR>
__global__ __launch_bounds__(256, 1) void skel(const char* __restrict__ src, float* out, unsigned long long* cyc, int iters, int ld_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x4 acc[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  i32x4 fa = {lane * 3 + 1, lane * 5 + 2, 0x3f803f80, 0x3f003f00};
  const char* base = src + (size_t)(blockIdx.x % 64) * 256 * ld_bytes;
  const uint32_t off_line = (uint32_t)((wave * 8 + (lane >> 3)) * ld_bytes + (lane & 7) * 16);
  constexpr bool READS = VAR != 2, DMA = VAR >= 2, BARRIER = VAR != 0 && VAR != 5, SPREAD = VAR == 4;
  constexpr int NDMA = VAR == 7 ? 8 : 16;
  // fragment read address of lane (fr, fq): row fr of a 128-byte-row image, 16-byte chunk (ks*4 + fq) ^ ((fr >> 1) & 7)
  const int fr = lane & 15, fq = lane >> 4;
  const uint32_t rd0 = (uint32_t)(((wave >> 1) * 128 + fr) * 128 + ((fq ^ ((fr >> 1) & 7)) << 4));
  const uint32_t rd1 = (uint32_t)(32768 + ((wave & 1) * 128 + fr) * 128 + ((fq ^ ((fr >> 1) & 7)) << 4));
  // two fragment sets, as in the GEMM: a block's MFMAs consume what the PREVIOUS block read (no read -> use dependency inside a block)
  u32x4 frag[16], frag2[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { frag[i] = u32x4{0x3f803f80u, (unsigned)lane, 0x3e803e80u, 0x3f803f80u}; frag2[i] = frag[i]; }
  for (int i = threadIdx.x; i < 32768; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0x3f803f80u;      // 128 KB of finite bf16
  __syncthreads();
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const uint32_t koff = (uint32_t)((it * 128) % 8192);
    const uint32_t st = (uint32_t)((it & 1) * 65536), sn = st ^ 65536u;
    // ---- block 1
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      if constexpr (READS)
        asm volatile("ds_read_b128 %0, %1" : "+v"(frag2[g]) : "v"((((g < 8 ? rd1 : rd0) + (uint32_t)((g & 7) * 2048)) ^ 64u) + st) : "memory");      // k-step 1: the other 64-byte half of the row
      if constexpr (DMA && SPREAD) {
        if (g % 2 == 0)
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + off_line + (uint32_t)((g / 2) * 32 * ld_bytes) + koff),
                                           (void __attribute__((address_space(3)))*)(smem + sn + (g / 2) * 4096 + wave * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) mfma_tied(acc[g * 4 + q], fa, __builtin_bit_cast(i32x4, frag[(g + q) & 15]));
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (READS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (DMA) {
      if constexpr (VAR == 6) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if constexpr (BARRIER) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); }
    __builtin_amdgcn_sched_barrier(0);
    // ---- block 2
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      if constexpr (DMA) {
        if (SPREAD ? (g % 2 == 0) : (g < NDMA))
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + off_line + (uint32_t)(((SPREAD ? 4 + g / 2 : g) & 7) * 32 * ld_bytes) + koff),
                                           (void __attribute__((address_space(3)))*)(smem + (SPREAD ? sn : st) + (SPREAD ? 8 + g / 2 : g) * 4096 + wave * 1024), 16, 0, 0);
      }
      if constexpr (READS)
        asm volatile("ds_read_b128 %0, %1" : "+v"(frag[g]) : "v"((g < 8 ? rd1 : rd0) + sn + (uint32_t)((g & 7) * 2048)) : "memory");
#pragma unroll
      for (int q = 0; q < 4; ++q) mfma_tied(acc[g * 4 + q], fa, __builtin_bit_cast(i32x4, frag2[(g + q) & 15]));
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (READS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) { asm volatile("" : "+a"(acc[i])); s += acc[i][0] + acc[i][3]; }
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int VAR>
double run_skel(const char* src, float* out, unsigned long long* cyc, int grid, int iters, int ld_bytes) {
  const int LDS = 128 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&skel<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  std::vector<unsigned long long> h(grid);
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((skel<VAR>), dim3(grid), dim3(256), LDS, 0, src, out, cyc, iters, ld_bytes);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    best = std::min(best, (double)h[grid / 2] / iters);
  }
  return best;
}

template <int KIND, int N>
double run(const char* src, float* out, unsigned long long* cyc, int grid, int iters, int ld_bytes) {
  const int LDS = 128 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<KIND, N>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  std::vector<unsigned long long> h(grid);
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((probe<KIND, N>), dim3(grid), dim3(256), LDS, 0, src, out, cyc, iters, ld_bytes);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    best = std::min(best, (double)h[grid / 2] / iters);
  }
  return best;
}

int main(int argc, char** argv) {
  // optional argument: row stride in bytes (default 8192 = K 4096 bf16; e.g. 8320 = rows padded by one 128-byte line, to tell
  // power-of-two-stride set conflicts from a per-request cost)
  const int grid = 256, iters = 256, ld_bytes = argc > 1 ? atoi(argv[1]) : 8192;
  if (ld_bytes < 8192 || ld_bytes % 128 != 0 || ld_bytes > 16384) { printf("row stride must be a multiple of 128 in [8192, 16384]\n"); return 1; }
  const size_t bytes = (size_t)64 * 256 * ld_bytes;               // 128 MB: 64 panels of 2 MB
  char* src; float* out; unsigned long long* cyc;
  hipMalloc(&src, bytes); hipMalloc(&out, grid * 256 * sizeof(float)); hipMalloc(&cyc, grid * sizeof(unsigned long long));
  std::vector<unsigned short> h(bytes / 2);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { printf("setup failed: %s\n", hipGetErrorString(e)); return 1; }
  printf("row stride %d bytes\n", ld_bytes);
  printf("cycles per iteration of 128 MFMA 16x16x32 (2048 matrix cycles), one wave per SIMD, all 256 CUs; N fetches per wave per iteration\n");
  printf("%-34s %8s %8s %8s %8s %8s\n", "kind", "N=0", "N=4", "N=8", "N=16", "N=32");
#define ROW(NAME, KIND)                                                                                             \
  {                                                                                                                 \
    const double c0 = run<K_NONE, 0>(src, out, cyc, grid, iters, ld_bytes);                                         \
    const double c4 = run<KIND, 4>(src, out, cyc, grid, iters, ld_bytes), c8 = run<KIND, 8>(src, out, cyc, grid, iters, ld_bytes); \
    const double c16 = run<KIND, 16>(src, out, cyc, grid, iters, ld_bytes), c32 = run<KIND, 32>(src, out, cyc, grid, iters, ld_bytes); \
    printf("%-34s %8.0f %8.0f %8.0f %8.0f %8.0f   per fetch (N=16): %.1f cycles\n", NAME, c0, c4, c8, c16, c32, (c16 - c0) / 16.0); \
    fflush(stdout);                                                                                                 \
  }
  ROW("global_load_lds_dwordx4 (8x128B)", K_GLDS)
  ROW("buffer_load_dwordx4 lds (8x128B)", K_BUFLDS)
  ROW("global_load_dwordx4 frag (16x64B)", K_GLOAD_FRAG)
  ROW("global_load_dwordx4 line (8x128B)", K_GLOAD_LINE)
  ROW("buffer_load_dwordx4 frag (16x64B)", K_BUFLOAD_FRAG)
  ROW("ds_read_b128", K_DSREAD)
  printf("\nGEMM K-loop skeleton (cycles per K-tile of 128 MFMA; 2048 = matrix pipe alone):\n");
  printf("  0 reads only (32 ds_read_b128)            %6.0f\n", run_skel<0>(src, out, cyc, grid, iters, ld_bytes));
  printf("  1 reads + barrier                         %6.0f\n", run_skel<1>(src, out, cyc, grid, iters, ld_bytes));
  printf("  2 DMA only (16, block 2) + vmcnt(0) + bar %6.0f\n", run_skel<2>(src, out, cyc, grid, iters, ld_bytes));
  printf("  3 reads + DMA + vmcnt(0) + barrier (GEMM) %6.0f\n", run_skel<3>(src, out, cyc, grid, iters, ld_bytes));
  printf("  4 like 3, DMA spread over both blocks     %6.0f\n", run_skel<4>(src, out, cyc, grid, iters, ld_bytes));
  printf("  5 like 3 without the barrier              %6.0f\n", run_skel<5>(src, out, cyc, grid, iters, ld_bytes));
  printf("  6 like 3, counted vmcnt(16)               %6.0f\n", run_skel<6>(src, out, cyc, grid, iters, ld_bytes));
  printf("  7 like 3, 8 DMA pieces                    %6.0f\n", run_skel<7>(src, out, cyc, grid, iters, ld_bytes));
  fflush(stdout);
  e = hipGetLastError();
  if (e != hipSuccess) { printf("run failed: %s\n", hipGetErrorString(e)); return 1; }
  return 0;
}
