// vlb_comm_*: the exchange steps of the sharded training step over RCCL / xGMI.
//
// An MI355X node is 8 GPUs fully connected by point-to-point xGMI links (7 links x ~153 GB/s per GPU).  A ring
// collective is bound by ONE link (7 hops); the "direct" schedules here move every shard over its own link at the
// same time: all-gather = every rank sends its shard to every peer (one grouped ncclSend/ncclRecv batch),
// reduce-scatter = every rank sends slice j of its buffer to rank j into a staging area, then ONE local kernel sums
// the `world` staged slices in rank order - a fixed order, so the reduced gradients are bit-reproducible whatever
// algorithm RCCL would have picked, and ranks that own different slices still agree on every shared scalar.
//
// RCCL is bound at first use with dlopen("librccl.so.1") - the same SONAME torch's bundled copy carries, so a
// process that already initialised torch.distributed shares ONE RCCL instance with libvlb - and libvlb.so itself
// keeps no link-time dependency on it (the CPU-side symbol tests load the library without RCCL present).
#include <dlfcn.h>

#include <mutex>

#include "common.hpp"

namespace {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclUint8 = 1, ncclFloat32 = 7, ncclBfloat16 = 9 };
enum { ncclSum = 0 };

struct Rccl {
  void* h = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.h) break;
  }
  if (!g_rccl.h) return;
#define VLB_SYM(field, name) *(void**)(&g_rccl.field) = dlsym(g_rccl.h, name); if (!g_rccl.field) return;
  VLB_SYM(GetUniqueId, "ncclGetUniqueId") VLB_SYM(CommInitRank, "ncclCommInitRank") VLB_SYM(CommDestroy, "ncclCommDestroy")
  VLB_SYM(GroupStart, "ncclGroupStart") VLB_SYM(GroupEnd, "ncclGroupEnd") VLB_SYM(Send, "ncclSend") VLB_SYM(Recv, "ncclRecv")
  VLB_SYM(AllReduce, "ncclAllReduce") VLB_SYM(GetErrorString, "ncclGetErrorString")
#undef VLB_SYM
  g_rccl.ok = true;
}

bool rccl_ready() {
  std::call_once(g_rccl_once, load_rccl);
  if (!g_rccl.ok) vlb_set_error("comm: librccl.so.1 could not be loaded (%s)", dlerror() ? dlerror() : "missing symbol");
  return g_rccl.ok;
}

struct VlbComm {
  ncclComm_t comm;
  int rank, world;
};

#define VLB_NCCL(call, what)                                                                 \
  do {                                                                                       \
    int r__ = (call);                                                                        \
    if (r__ != ncclSuccess) {                                                                \
      vlb_set_error("comm: %s failed: %s", what, g_rccl.GetErrorString(r__));                \
      return VLB_ERR_LAUNCH;                                                                 \
    }                                                                                        \
  } while (0)

// out[i] = sum over r = 0..world-1 (in that order) of stage[r*n + i]; fp32, 16 B per lane
__global__ __launch_bounds__(256) void reduce_slices_kernel(const float* __restrict__ stage, float* __restrict__ out, int64_t n4,
                                                            int64_t n, int world) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  f32x4 acc = *reinterpret_cast<const f32x4*>(stage + i * 4);
  for (int r = 1; r < world; ++r) acc += *reinterpret_cast<const f32x4*>(stage + (int64_t)r * n + i * 4);
  *reinterpret_cast<f32x4*>(out + i * 4) = acc;
}

// bf16 form: 8 elements per lane; the `world` staged slices are widened to fp32, summed in rank order and rounded to
// bf16 ONCE (the bf16 gradients of the full fine-tune's backbone store: what a bf16 ncclSum reduce-scatter would round
// after every hop is rounded here a single time).
__global__ __launch_bounds__(256) void reduce_slices_bf16_kernel(const bf16* __restrict__ stage, bf16* __restrict__ out, int64_t n8,
                                                                 int64_t n, int world) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float acc[8];
  bf16x8 v = *reinterpret_cast<const bf16x8*>(stage + i * 8);
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = (float)v[j];
  for (int r = 1; r < world; ++r) {
    v = *reinterpret_cast<const bf16x8*>(stage + (int64_t)r * n + i * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (bf16)acc[j];
  *reinterpret_cast<bf16x8*>(out + i * 8) = v;
}

int launch_reduce_slices(const void* stage, void* out, int64_t n, int world, int is_bf16, hipStream_t st) {
  const int64_t per = is_bf16 ? 8 : 4, nv = n / per;
  const dim3 grid((unsigned)((nv + 255) / 256));
  if (is_bf16)
    hipLaunchKernelGGL(reduce_slices_bf16_kernel, grid, dim3(256), 0, st, static_cast<const bf16*>(stage), static_cast<bf16*>(out), nv, n, world);
  else
    hipLaunchKernelGGL(reduce_slices_kernel, grid, dim3(256), 0, st, static_cast<const float*>(stage), static_cast<float*>(out), nv, n, world);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

// the exchange of a direct reduce-scatter: slice r of `stage` <- rank r's slice `rank` of ITS `send` buffer
int exchange_slices(VlbComm* c, const char* send, char* stage, int64_t slice_bytes, hipStream_t st) {
  hipError_t e = hipMemcpyAsync(stage + (int64_t)c->rank * slice_bytes, send + (int64_t)c->rank * slice_bytes, (size_t)slice_bytes,
                                hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) { vlb_set_error("reducescatter_direct: local copy failed: %s", hipGetErrorString(e)); return VLB_ERR_LAUNCH; }
  if (c->world > 1) {
    VLB_NCCL(g_rccl.GroupStart(), "ncclGroupStart");
    for (int p = 0; p < c->world; ++p) {
      if (p == c->rank) continue;
      VLB_NCCL(g_rccl.Send(send + (int64_t)p * slice_bytes, (size_t)slice_bytes, ncclUint8, p, c->comm, st), "ncclSend");
      VLB_NCCL(g_rccl.Recv(stage + (int64_t)p * slice_bytes, (size_t)slice_bytes, ncclUint8, p, c->comm, st), "ncclRecv");
    }
    VLB_NCCL(g_rccl.GroupEnd(), "ncclGroupEnd");
  }
  return VLB_OK;
}
}  // namespace

extern "C" int vlb_comm_unique_id(void* id128_host) {
  VLB_REQUIRE(id128_host, "comm_unique_id: null buffer");
  if (!rccl_ready()) return VLB_ERR_LAUNCH;
  VLB_NCCL(g_rccl.GetUniqueId(reinterpret_cast<ncclUniqueId*>(id128_host)), "ncclGetUniqueId");
  return VLB_OK;
}

extern "C" int vlb_comm_init(int rank, int world, const void* id128_host, void** comm_out) {
  VLB_REQUIRE(comm_out && id128_host && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments (rank %d of %d)", rank, world);
  if (!rccl_ready()) return VLB_ERR_LAUNCH;
  ncclUniqueId id;
  memcpy(&id, id128_host, sizeof(id));
  VlbComm* c = new VlbComm{nullptr, rank, world};
  int r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    vlb_set_error("comm_init: ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
    delete c;
    return VLB_ERR_LAUNCH;
  }
  *comm_out = c;
  return VLB_OK;
}

extern "C" int vlb_comm_destroy(void* comm) {
  if (!comm) return VLB_OK;
  VlbComm* c = static_cast<VlbComm*>(comm);
  if (g_rccl.ok && c->comm) g_rccl.CommDestroy(c->comm);
  delete c;
  return VLB_OK;
}

extern "C" int vlb_comm_rank(void* comm) { return comm ? static_cast<VlbComm*>(comm)->rank : -1; }
extern "C" int vlb_comm_world(void* comm) { return comm ? static_cast<VlbComm*>(comm)->world : -1; }

extern "C" int vlb_allgather_direct(void* comm, const void* shard, void* full, int64_t shard_bytes, void* stream) {
  VLB_REQUIRE(comm && shard && full && shard_bytes > 0, "allgather_direct: bad arguments");
  VlbComm* c = static_cast<VlbComm*>(comm);
  hipStream_t st = as_stream(stream);
  char* dst = static_cast<char*>(full);
  if (dst + (int64_t)c->rank * shard_bytes != shard) {      // own shard: local copy (a no-op when gathered in place)
    hipError_t e = hipMemcpyAsync(dst + (int64_t)c->rank * shard_bytes, shard, (size_t)shard_bytes, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) { vlb_set_error("allgather_direct: local copy failed: %s", hipGetErrorString(e)); return VLB_ERR_LAUNCH; }
  }
  if (c->world == 1) return VLB_OK;
  VLB_NCCL(g_rccl.GroupStart(), "ncclGroupStart");
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank) continue;
    VLB_NCCL(g_rccl.Send(shard, (size_t)shard_bytes, ncclUint8, p, c->comm, st), "ncclSend");
    VLB_NCCL(g_rccl.Recv(dst + (int64_t)p * shard_bytes, (size_t)shard_bytes, ncclUint8, p, c->comm, st), "ncclRecv");
  }
  VLB_NCCL(g_rccl.GroupEnd(), "ncclGroupEnd");
  return VLB_OK;
}

extern "C" int64_t vlb_reducescatter_stage_floats(int64_t n_per_rank, int world) { return n_per_rank * (int64_t)world; }

extern "C" int vlb_reduce_slices(const void* stage, void* out, int64_t n_per_rank, int world, int is_bf16, void* stream) {
  const int per = is_bf16 ? 8 : 4;
  VLB_REQUIRE(stage && out && n_per_rank > 0 && world >= 1 && n_per_rank % per == 0,
              "reduce_slices: bad arguments (n_per_rank must be a multiple of %d)", per);
  VLB_REQUIRE((((uintptr_t)stage | (uintptr_t)out) % 16) == 0 && ((n_per_rank * (is_bf16 ? 2 : 4)) % 16) == 0,
              "reduce_slices: buffers and slices must be 16-byte aligned");
  return launch_reduce_slices(stage, out, n_per_rank, world, is_bf16, as_stream(stream));
}

extern "C" int vlb_reducescatter_direct(void* comm, const float* send, float* out, int64_t n_per_rank, float* stage, void* stream) {
  VLB_REQUIRE(comm && send && out && stage && n_per_rank > 0 && n_per_rank % 4 == 0, "reducescatter_direct: bad arguments (n_per_rank must be a multiple of 4)");
  VLB_REQUIRE((((uintptr_t)send | (uintptr_t)out | (uintptr_t)stage) % 16) == 0, "reducescatter_direct: buffers must be 16-byte aligned");
  VlbComm* c = static_cast<VlbComm*>(comm);
  hipStream_t st = as_stream(stream);
  int rc = exchange_slices(c, reinterpret_cast<const char*>(send), reinterpret_cast<char*>(stage), n_per_rank * (int64_t)sizeof(float), st);
  if (rc != VLB_OK) return rc;
  return launch_reduce_slices(stage, out, n_per_rank, c->world, 0, st);
}

extern "C" int vlb_reducescatter_direct_bf16(void* comm, const void* send, void* out, int64_t n_per_rank, void* stage, void* stream) {
  VLB_REQUIRE(comm && send && out && stage && n_per_rank > 0 && n_per_rank % 8 == 0, "reducescatter_direct_bf16: bad arguments (n_per_rank must be a multiple of 8)");
  VLB_REQUIRE((((uintptr_t)send | (uintptr_t)out | (uintptr_t)stage) % 16) == 0, "reducescatter_direct_bf16: buffers must be 16-byte aligned");
  VlbComm* c = static_cast<VlbComm*>(comm);
  hipStream_t st = as_stream(stream);
  int rc = exchange_slices(c, static_cast<const char*>(send), static_cast<char*>(stage), n_per_rank * 2, st);
  if (rc != VLB_OK) return rc;
  return launch_reduce_slices(stage, out, n_per_rank, c->world, 1, st);
}

extern "C" int vlb_allreduce_scalar(void* comm, float* values, int count, void* stream) {
  VLB_REQUIRE(comm && values && count > 0, "allreduce_scalar: bad arguments");
  VlbComm* c = static_cast<VlbComm*>(comm);
  if (c->world == 1) return VLB_OK;
  VLB_NCCL(g_rccl.AllReduce(values, values, (size_t)count, ncclFloat32, ncclSum, c->comm, as_stream(stream)), "ncclAllReduce");
  return VLB_OK;
}
