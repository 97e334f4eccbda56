// zk_comm.hip -- RCCL through the C-ABI (SURVEY.md 8(b) last row, 8(e)).
//
// Replaces the cross-device copies behind utils/parallel.py:134-208 (average_gradients: per-variable
// concat + reduce_mean over the towers) with sum collectives over xGMI on a HIP stream of the caller's
// choice (the gradient side stream, so that a bucket's exchange overlaps the rest of the backward); the
// 1/N of the mean is folded into the optimiser pass (zk_adam_step's grad_scale).
//
// Ownership: the communicator is created here and returned as an opaque handle that the CALLER owns and
// must hand back to zk_comm_destroy; the library keeps no global communicator state.  The rendezvous (who is
// rank 0, how the 128-byte unique id reaches the other ranks) is the caller's: zero_amd/utils/parallel.py sends
// it through the torch.distributed store.
//
// librccl is resolved with dlopen at the first call, not at link time: libzero_hip.so must load on a box
// without RCCL (the CPU-side ABI test), and inside a PyTorch process the already loaded librccl is reused.
#include "zk_common.h"
#include <dlfcn.h>
#include <string.h>

// ---- stream hand-off without an event on the compute stream (round 6) -----------------------------------------------
// The reducer orders RCCL's stream behind the backward once per gradient bucket.  An event recorded on the compute stream
// did that -- and the single-rank loop showed what such an event costs there: every dispatch of the stream a little
// slower, 40-65 us per step over ~145 launches.  The alternative: a monotonic 64-bit word in device memory.
//   zk_flag_add   (compute stream)  one thread: the word += 1, released at agent scope -- everything enqueued on that
//                                   stream before it is complete and visible
//   zk_flag_wait  (RCCL's stream)   one thread polls until the word >= target (acquire), then ends: the collective
//                                   enqueued behind it starts.  Bounded: after ~2 s without progress it raises the
//                                   error word (if given) and ends, so that a lost hand-off is a reported error, not a hang.
__global__ void k_flag_add(unsigned long long* flag) {
  __hip_atomic_fetch_add(flag, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_flag_wait(const unsigned long long* flag, unsigned long long target, int* err) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();           // 100 MHz
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(32);
    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
      if (err != nullptr) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
}

namespace {
struct Uid { char internal[128]; };
struct Rccl {
  void* h = nullptr;
  int (*GetUniqueId)(Uid*) = nullptr;
  int (*CommInitRank)(void**, int, Uid, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  const char* err = nullptr;
};
Rccl g_rccl;

const Rccl* rccl() {
  if (g_rccl.h != nullptr || g_rccl.err != nullptr) return &g_rccl;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h != nullptr) break;
  }
  if (h == nullptr) { g_rccl.err = "librccl.so not found (dlopen)"; return &g_rccl; }
#define ZK_SYM(field, name)                                                     \
  *(void**)(&g_rccl.field) = dlsym(h, name);                                    \
  if (g_rccl.field == nullptr) { g_rccl.err = "librccl: missing symbol " name; return &g_rccl; }
  ZK_SYM(GetUniqueId, "ncclGetUniqueId")
  ZK_SYM(CommInitRank, "ncclCommInitRank")
  ZK_SYM(CommDestroy, "ncclCommDestroy")
  ZK_SYM(AllReduce, "ncclAllReduce")
  ZK_SYM(ReduceScatter, "ncclReduceScatter")
  ZK_SYM(AllGather, "ncclAllGather")
  ZK_SYM(GroupStart, "ncclGroupStart")
  ZK_SYM(GroupEnd, "ncclGroupEnd")
  ZK_SYM(GetErrorString, "ncclGetErrorString")
#undef ZK_SYM
  g_rccl.h = h;
  return &g_rccl;
}

struct Comm { void* nccl; int nranks, rank; };

// dtype codes of the C-ABI: 0 = fp32, 1 = bf16 (ncclFloat32 = 7, ncclBfloat16 = 9); op is always sum (0)
int nccl_dtype(int dtype) { return dtype == 1 ? 9 : 7; }

int fail(const Rccl* r, int rc, const char* what) {
  // ncclResult_t values are small positive integers; offset them so they cannot be read as a hipError_t
  return zk_set_error(1000 + rc, "%s: %s", what, r->GetErrorString ? r->GetErrorString(rc) : "rccl error");
}
}  // namespace

extern "C" {
int zk_flag_add(void* flag, hipStream_t stream) {
  ZK_CHECK_ARG(flag != nullptr && (((uintptr_t)flag) & 7) == 0, "zk_flag_add: an 8-byte aligned device word is required");
  hipLaunchKernelGGL(k_flag_add, dim3(1), dim3(1), 0, stream, (unsigned long long*)flag);
  ZK_LAUNCH_CHECK();
  return 0;
}
int zk_flag_wait(const void* flag, unsigned long long target, int* err, hipStream_t stream) {
  ZK_CHECK_ARG(flag != nullptr && (((uintptr_t)flag) & 7) == 0, "zk_flag_wait: an 8-byte aligned device word is required");
  hipLaunchKernelGGL(k_flag_wait, dim3(1), dim3(1), 0, stream, (const unsigned long long*)flag, target, err);
  ZK_LAUNCH_CHECK();
  return 0;
}
// 1 when librccl could be loaded, 0 otherwise (message in zk_last_error_string)
int zk_comm_available(void) {
  const Rccl* r = rccl();
  if (r->err != nullptr) { zk_set_error(-1, "%s", r->err); return 0; }
  return 1;
}
// rank 0 only: fill the 128-byte id every rank must pass to zk_comm_init
int zk_comm_unique_id(void* id128) {
  const Rccl* r = rccl();
  ZK_CHECK_ARG(r->err == nullptr, "zk_comm_unique_id: %s", r->err);
  ZK_CHECK_ARG(id128 != nullptr, "zk_comm_unique_id: null buffer");
  Uid u;
  const int rc = r->GetUniqueId(&u);
  if (rc != 0) return fail(r, rc, "ncclGetUniqueId");
  memcpy(id128, u.internal, 128);
  return 0;
}
// collective over all ranks (blocks until every rank has called it); the current HIP device is the rank's GPU
int zk_comm_init(const void* id128, int nranks, int rank, void** comm_out) {
  const Rccl* r = rccl();
  ZK_CHECK_ARG(r->err == nullptr, "zk_comm_init: %s", r->err);
  ZK_CHECK_ARG(id128 != nullptr && comm_out != nullptr, "zk_comm_init: null argument");
  ZK_CHECK_ARG(nranks >= 1 && rank >= 0 && rank < nranks, "zk_comm_init: rank %d of %d", rank, nranks);
  Uid u;
  memcpy(u.internal, id128, 128);
  void* c = nullptr;
  const int rc = r->CommInitRank(&c, nranks, u, rank);
  if (rc != 0) return fail(r, rc, "ncclCommInitRank");
  Comm* h = new Comm{c, nranks, rank};
  *comm_out = h;
  return 0;
}
int zk_comm_destroy(void* comm) {
  if (comm == nullptr) return 0;
  const Rccl* r = rccl();
  Comm* h = (Comm*)comm;
  int rc = 0;
  if (r->err == nullptr && h->nccl != nullptr) rc = r->CommDestroy(h->nccl);
  delete h;
  if (rc != 0) return fail(r, rc, "ncclCommDestroy");
  return 0;
}
int zk_comm_size(const void* comm) { return comm ? ((const Comm*)comm)->nranks : 0; }

// in-place sum all-reduce of `count` elements (dtype 0 = fp32, 1 = bf16) enqueued on `stream`
int zk_comm_allreduce(void* comm, void* buf, size_t count, int dtype, hipStream_t stream) {
  const Rccl* r = rccl();
  ZK_CHECK_ARG(r->err == nullptr && comm != nullptr, "zk_comm_allreduce: no communicator");
  ZK_CHECK_ARG(dtype == 0 || dtype == 1, "zk_comm_allreduce: dtype must be 0 (fp32) or 1 (bf16)");
  if (count == 0) return 0;
  const int rc = r->AllReduce(buf, buf, count, nccl_dtype(dtype), 0, ((Comm*)comm)->nccl, stream);
  if (rc != 0) return fail(r, rc, "ncclAllReduce");
  return 0;
}
// several in-place sum all-reduces fused into ONE RCCL group (one launch on the stream): the small buckets of a
// layer group.  bufs / counts: host arrays of n entries.
int zk_comm_allreduce_multi(void* comm, void* const* bufs, const size_t* counts, int n, int dtype,
                            hipStream_t stream) {
  const Rccl* r = rccl();
  ZK_CHECK_ARG(r->err == nullptr && comm != nullptr, "zk_comm_allreduce_multi: no communicator");
  ZK_CHECK_ARG(dtype == 0 || dtype == 1, "zk_comm_allreduce_multi: dtype must be 0 (fp32) or 1 (bf16)");
  ZK_CHECK_ARG(n >= 0 && (n == 0 || (bufs != nullptr && counts != nullptr)), "zk_comm_allreduce_multi: bad list");
  int rc = r->GroupStart();
  if (rc != 0) return fail(r, rc, "ncclGroupStart");
  for (int i = 0; i < n && rc == 0; ++i)
    if (counts[i]) rc = r->AllReduce(bufs[i], bufs[i], counts[i], nccl_dtype(dtype), 0, ((Comm*)comm)->nccl, stream);
  const int rc2 = r->GroupEnd();
  if (rc != 0) return fail(r, rc, "ncclAllReduce (group)");
  if (rc2 != 0) return fail(r, rc2, "ncclGroupEnd");
  return 0;
}
// recv[rank*count .. +count) <- send of every rank (the touched rows of the source-embedding gradient,
// utils/parallel.py:142-181 concatenates the towers' IndexedSlices before deduplicating them)
int zk_comm_allgather(void* comm, const void* send, void* recv, size_t count, int dtype, hipStream_t stream) {
  const Rccl* r = rccl();
  ZK_CHECK_ARG(r->err == nullptr && comm != nullptr, "zk_comm_allgather: no communicator");
  ZK_CHECK_ARG(dtype == 0 || dtype == 1 || dtype == 2, "zk_comm_allgather: dtype must be 0 (fp32), 1 (bf16) or 2 (int32)");
  if (count == 0) return 0;
  const int dt = dtype == 2 ? 2 /* ncclInt32 */ : nccl_dtype(dtype);
  const int rc = r->AllGather(send, recv, count, dt, ((Comm*)comm)->nccl, stream);
  if (rc != 0) return fail(r, rc, "ncclAllGather");
  return 0;
}
}  // extern "C"
