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

extern "C" int vlb_reducescatter_direct(void* comm, const float* send, float* out, int64_t n_per_rank, float* stage, void* stream) {
  VLB_REQUIRE(comm && send && out && stage && n_per_rank > 0 && n_per_rank % 4 == 0, "reducescatter_direct: bad arguments (n_per_rank must be a multiple of 4)");
  VLB_REQUIRE((((uintptr_t)send | (uintptr_t)out | (uintptr_t)stage) % 16) == 0, "reducescatter_direct: buffers must be 16-byte aligned");
  VlbComm* c = static_cast<VlbComm*>(comm);
  hipStream_t st = as_stream(stream);
  const int64_t n = n_per_rank;
  // slice r of the staging area <- rank r's slice `rank` of ITS buffer (own slice: local copy)
  hipError_t e = hipMemcpyAsync(stage + (int64_t)c->rank * n, send + (int64_t)c->rank * n, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) { vlb_set_error("reducescatter_direct: local copy failed: %s", hipGetErrorString(e)); return VLB_ERR_LAUNCH; }
  if (c->world > 1) {
    VLB_NCCL(g_rccl.GroupStart(), "ncclGroupStart");
    for (int p = 0; p < c->world; ++p) {
      if (p == c->rank) continue;
      VLB_NCCL(g_rccl.Send(send + (int64_t)p * n, (size_t)n, ncclFloat32, p, c->comm, st), "ncclSend");
      VLB_NCCL(g_rccl.Recv(stage + (int64_t)p * n, (size_t)n, ncclFloat32, p, c->comm, st), "ncclRecv");
    }
    VLB_NCCL(g_rccl.GroupEnd(), "ncclGroupEnd");
  }
  const int64_t n4 = n / 4;
  hipLaunchKernelGGL(reduce_slices_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, stage, out, n4, n, c->world);
  VLB_LAUNCH_CHECK();
  return VLB_OK;
}

extern "C" int vlb_allreduce_scalar(void* comm, float* values, int count, void* stream) {
  VLB_REQUIRE(comm && values && count > 0, "allreduce_scalar: bad arguments");
  VlbComm* c = static_cast<VlbComm*>(comm);
  if (c->world == 1) return VLB_OK;
  VLB_NCCL(g_rccl.AllReduce(values, values, (size_t)count, ncclFloat32, ncclSum, c->comm, as_stream(stream)), "ncclAllReduce");
  return VLB_OK;
}
