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

#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <vector>

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

// ---- in-process loopback transport -------------------------------------------------------------------------------
// `world` communicators in ONE process whose sends and receives pair up through device copies (vlb_comm_init_loopback):
// the exchange schedules below - peer loops, slice offsets, staging layout, the rank-ordered reduction - then run with
// world 2..8 on a single GPU, every "rank" driven by its own host thread and stream, exactly as `world` processes drive
// them over RCCL.  Semantics kept from ncclSend / ncclRecv inside a group: nothing moves before GroupEnd; GroupEnd returns
// once every operation of the group is ENQUEUED in stream order - the receiver's stream waits for the sender's stream (the
// send buffer as of the sender's GroupEnd), copies, and the sender's stream then waits for that copy (later writes to the
// send buffer are ordered behind it).  Matching is FIFO per (source, destination) pair.  A peer that never arrives is an
// error after LOOP_TIMEOUT_S, not a hang.
constexpr int LOOP_TIMEOUT_S = 60;
constexpr int LOOP_AR_MAX = 256;            // floats per vlb_allreduce_scalar call on the loopback transport
struct LoopOp {
  const void* src = nullptr; void* dst = nullptr; size_t bytes = 0;
  hipEvent_t ready = nullptr, done = nullptr;
  bool has_ready = false, has_done = false;
};
struct LoopHub {
  int world, refs;
  std::mutex mu;
  std::condition_variable cv;
  std::map<std::tuple<int, int, uint64_t>, std::shared_ptr<LoopOp>> ops;      // (source, destination, sequence number)
  std::vector<std::vector<uint64_t>> send_seq, recv_seq;
  // all-reduce: [parity][rank][LOOP_AR_MAX] floats of caller-owned device memory + per-(round, rank) "staged" events
  float* stage;
  std::vector<uint64_t> ar_round;
  std::map<std::pair<uint64_t, int>, hipEvent_t> ar_ready;
  LoopHub(int w, float* st) : world(w), refs(w), send_seq(w, std::vector<uint64_t>(w, 0)), recv_seq(w, std::vector<uint64_t>(w, 0)), stage(st), ar_round(w, 0) {}
};

struct VlbComm {
  ncclComm_t comm;
  int rank, world;
  LoopHub* hub = nullptr;                                   // non-null: loopback transport instead of RCCL
  std::vector<std::shared_ptr<LoopOp>> pend_send, pend_recv;          // this rank's open group
  std::vector<std::tuple<int, int, uint64_t>> pend_keys;              // map keys of pend_send (erased by the sender)
};

#define VLB_NCCL(call, what)                                                                 \
  do {                                                                                       \
    int r__ = (call);                                                                        \
    if (r__ != ncclSuccess) {                                                                \
      vlb_set_error("comm: %s failed: %s", what, g_rccl.GetErrorString(r__));                \
      return VLB_ERR_LAUNCH;                                                                 \
    }                                                                                        \
  } while (0)

#define VLB_HIP(call, what)                                                                  \
  do {                                                                                       \
    hipError_t e__ = (call);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      vlb_set_error("comm (loopback): %s failed: %s", what, hipGetErrorString(e__));         \
      return VLB_ERR_LAUNCH;                                                                 \
    }                                                                                        \
  } while (0)

// ---- transport: RCCL, or the loopback hub -----------------------------------------------------------------------
int t_group_start(VlbComm* c) {
  if (!c->hub) VLB_NCCL(g_rccl.GroupStart(), "ncclGroupStart");
  return VLB_OK;
}
int t_send(VlbComm* c, const void* buf, size_t bytes, int peer, hipStream_t st) {
  if (!c->hub) { VLB_NCCL(g_rccl.Send(buf, bytes, ncclUint8, peer, c->comm, st), "ncclSend"); return VLB_OK; }
  LoopHub* h = c->hub;
  std::lock_guard<std::mutex> lk(h->mu);
  const auto key = std::make_tuple(c->rank, peer, h->send_seq[c->rank][peer]++);
  auto& op = h->ops[key];
  if (!op) op = std::make_shared<LoopOp>();
  op->src = buf; op->bytes = bytes;
  c->pend_send.push_back(op);
  c->pend_keys.push_back(key);
  return VLB_OK;
}
int t_recv(VlbComm* c, void* buf, size_t bytes, int peer, hipStream_t st) {
  if (!c->hub) { VLB_NCCL(g_rccl.Recv(buf, bytes, ncclUint8, peer, c->comm, st), "ncclRecv"); return VLB_OK; }
  LoopHub* h = c->hub;
  std::lock_guard<std::mutex> lk(h->mu);
  auto& op = h->ops[std::make_tuple(peer, c->rank, h->recv_seq[peer][c->rank]++)];
  if (!op) op = std::make_shared<LoopOp>();
  op->dst = buf;
  if (op->bytes == 0) op->bytes = bytes;
  c->pend_recv.push_back(op);
  return VLB_OK;
}
int t_group_end(VlbComm* c, hipStream_t st) {
  if (!c->hub) { VLB_NCCL(g_rccl.GroupEnd(), "ncclGroupEnd"); return VLB_OK; }
  LoopHub* h = c->hub;
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(LOOP_TIMEOUT_S);
  for (auto& op : c->pend_send) {                                     // 1. my send buffers are final from here on (stream order)
    hipEvent_t ev;
    VLB_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
    VLB_HIP(hipEventRecord(ev, st), "hipEventRecord");
    std::lock_guard<std::mutex> lk(h->mu);
    op->ready = ev; op->has_ready = true;
  }
  h->cv.notify_all();
  for (auto& op : c->pend_recv) {                                     // 2. my receives: behind the sender's stream, then copy
    {
      std::unique_lock<std::mutex> lk(h->mu);
      if (!h->cv.wait_until(lk, deadline, [&] { return op->has_ready; })) {
        vlb_set_error("comm (loopback): rank %d waited %d s for a peer's send (every rank must be driven by its own thread)", c->rank, LOOP_TIMEOUT_S);
        return VLB_ERR_LAUNCH;
      }
    }
    VLB_HIP(hipStreamWaitEvent(st, op->ready, 0), "hipStreamWaitEvent");
    VLB_HIP(hipMemcpyAsync(op->dst, op->src, op->bytes, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync");
    hipEvent_t ev;
    VLB_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
    VLB_HIP(hipEventRecord(ev, st), "hipEventRecord");
    std::lock_guard<std::mutex> lk(h->mu);
    op->done = ev; op->has_done = true;
    h->cv.notify_all();
  }
  for (size_t i = 0; i < c->pend_send.size(); ++i) {                  // 3. my sends: my stream continues behind the peers' copies
    auto& op = c->pend_send[i];
    {
      std::unique_lock<std::mutex> lk(h->mu);
      if (!h->cv.wait_until(lk, deadline, [&] { return op->has_done; })) {
        vlb_set_error("comm (loopback): rank %d waited %d s for a peer's receive", c->rank, LOOP_TIMEOUT_S);
        return VLB_ERR_LAUNCH;
      }
      h->ops.erase(c->pend_keys[i]);
    }
    VLB_HIP(hipStreamWaitEvent(st, op->done, 0), "hipStreamWaitEvent");
    (void)hipEventDestroy(op->ready);                                  // (released once the enqueued waits have passed)
    (void)hipEventDestroy(op->done);
  }
  c->pend_send.clear(); c->pend_recv.clear(); c->pend_keys.clear();
  return VLB_OK;
}

// loopback all-reduce (sum): every rank stages its values, then sums all ranks' staged values in rank order
__global__ void loop_allreduce_kernel(const float* __restrict__ stage, float* __restrict__ out, int count, int world) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float acc = 0.f;
  for (int r = 0; r < world; ++r) acc += stage[r * LOOP_AR_MAX + i];
  out[i] = acc;
}
int loop_allreduce(VlbComm* c, float* values, int count, hipStream_t st) {
  LoopHub* h = c->hub;
  if (count > LOOP_AR_MAX) { vlb_set_error("comm (loopback): allreduce_scalar of %d > %d values", count, LOOP_AR_MAX); return VLB_ERR_INVALID; }
  uint64_t round;
  { std::lock_guard<std::mutex> lk(h->mu); round = h->ar_round[c->rank]++; }
  float* slab = h->stage + (size_t)(round & 1) * h->world * LOOP_AR_MAX;
  VLB_HIP(hipMemcpyAsync(slab + (size_t)c->rank * LOOP_AR_MAX, values, (size_t)count * sizeof(float), hipMemcpyDeviceToDevice, st), "hipMemcpyAsync");
  hipEvent_t ev;
  VLB_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
  VLB_HIP(hipEventRecord(ev, st), "hipEventRecord");
  { std::lock_guard<std::mutex> lk(h->mu); h->ar_ready[{round, c->rank}] = ev; }
  h->cv.notify_all();
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(LOOP_TIMEOUT_S);
  for (int r = 0; r < h->world; ++r) {
    hipEvent_t pe;
    {
      std::unique_lock<std::mutex> lk(h->mu);
      if (!h->cv.wait_until(lk, deadline, [&] { return h->ar_ready.count({round, r}) != 0; })) {
        vlb_set_error("comm (loopback): rank %d waited %d s for rank %d in an all-reduce", c->rank, LOOP_TIMEOUT_S, r);
        return VLB_ERR_LAUNCH;
      }
      pe = h->ar_ready[{round, r}];
    }
    VLB_HIP(hipStreamWaitEvent(st, pe, 0), "hipStreamWaitEvent");
  }
  hipLaunchKernelGGL(loop_allreduce_kernel, dim3((count + 63) / 64), dim3(64), 0, st, slab, values, count, h->world);
  VLB_LAUNCH_CHECK();
  // round - 2's events are no longer needed by anyone (every rank has entered round - 1, hence waited on them)
  if (round >= 2) {
    std::lock_guard<std::mutex> lk(h->mu);
    auto it = h->ar_ready.find({round - 2, c->rank});
    if (it != h->ar_ready.end()) { (void)hipEventDestroy(it->second); h->ar_ready.erase(it); }
  }
  return VLB_OK;
}

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
    int rc = t_group_start(c);
    if (rc != VLB_OK) return rc;
    for (int p = 0; p < c->world; ++p) {
      if (p == c->rank) continue;
      if ((rc = t_send(c, send + (int64_t)p * slice_bytes, (size_t)slice_bytes, p, st)) != VLB_OK) return rc;
      if ((rc = t_recv(c, stage + (int64_t)p * slice_bytes, (size_t)slice_bytes, p, st)) != VLB_OK) return rc;
    }
    return t_group_end(c, st);
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
  VlbComm* c = new VlbComm();
  c->comm = nullptr; c->rank = rank; c->world = world;
  int r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    vlb_set_error("comm_init: ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
    delete c;
    return VLB_ERR_LAUNCH;
  }
  *comm_out = c;
  return VLB_OK;
}

extern "C" int64_t vlb_comm_loopback_stage_bytes(int world) { return (int64_t)2 * world * LOOP_AR_MAX * (int64_t)sizeof(float); }

extern "C" int vlb_comm_init_loopback(int world, void* stage, int64_t stage_bytes, void** comms_out) {
  VLB_REQUIRE(comms_out && world >= 1 && world <= 64, "comm_init_loopback: bad arguments (world %d)", world);
  VLB_REQUIRE(stage && stage_bytes >= vlb_comm_loopback_stage_bytes(world) && ((uintptr_t)stage % 16) == 0,
              "comm_init_loopback: the all-reduce staging buffer must hold vlb_comm_loopback_stage_bytes(world) bytes");
  LoopHub* h = new LoopHub(world, static_cast<float*>(stage));
  for (int r = 0; r < world; ++r) {
    VlbComm* c = new VlbComm();
    c->comm = nullptr; c->rank = r; c->world = world; c->hub = h;
    comms_out[r] = c;
  }
  return VLB_OK;
}

extern "C" int vlb_comm_destroy(void* comm) {
  if (!comm) return VLB_OK;
  VlbComm* c = static_cast<VlbComm*>(comm);
  if (c->hub) {
    bool last;
    { std::lock_guard<std::mutex> lk(c->hub->mu); last = --c->hub->refs == 0; }
    if (last) {
      for (auto& kv : c->hub->ar_ready) (void)hipEventDestroy(kv.second);
      delete c->hub;
    }
  } else if (g_rccl.ok && c->comm) {
    g_rccl.CommDestroy(c->comm);
  }
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
  int rc = t_group_start(c);
  if (rc != VLB_OK) return rc;
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank) continue;
    if ((rc = t_send(c, shard, (size_t)shard_bytes, p, st)) != VLB_OK) return rc;
    if ((rc = t_recv(c, dst + (int64_t)p * shard_bytes, (size_t)shard_bytes, p, st)) != VLB_OK) return rc;
  }
  return t_group_end(c, st);
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
  if (c->hub) return loop_allreduce(c, values, count, as_stream(stream));
  VLB_NCCL(g_rccl.AllReduce(values, values, (size_t)count, ncclFloat32, ncclSum, c->comm, as_stream(stream)), "ncclAllReduce");
  return VLB_OK;
}
