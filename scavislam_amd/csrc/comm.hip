// comm.hip -- the library-owned collective of the landmark-sharded back-end: one RCCL communicator per
// context, ncclAllReduce(ncclDouble, ncclSum) of the packed reduced camera system on the context's stream, called from
// inside svs_ba_optimize -- no host callback, so a C++ host (stereo_slam's backend thread) gets the multi-GPU path too.
// The reference has no distributed code; this is the north_star's "RCCL all-reduce of the reduced camera
// Hessian over xGMI".  librccl is bound at run time (dlopen), so the single-GPU library carries no dependency on it and a
// process that already holds an RCCL instance (PyTorch ships one) shares it.
#include "common.h"
#include <dlfcn.h>
#include <string.h>
// The handful of RCCL declarations this file needs, written out (the library is bound at run time with dlopen; building the single-GPU product must
// not require the rccl development headers).  Values are the stable NCCL 2.x ABI: ncclFloat64 = 8, ncclSum = 0, a 128-byte unique id.
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
}

// ---- second transport: one-shot exchange over peer-mapped mailboxes (xGMI P2P), no RCCL -------------------------------------------------------------------
// The messages of the sharded back end are small (reduced system of a 50-pose window: ~150 KB; the trial scalars: 256 B) and latency-bound; a ring
// all-reduce pays 2 (N-1) hops for them.  One-shot: every rank PUSHES its vector into slot [rank] of every peer's mailbox (N-1 direct xGMI writes, all links
// at once), raises a flag there, waits for the N-1 flags in its own mailbox and sums the N vectors in rank order 0..N-1 -- one hop, and every rank adds the
// same numbers in the same order, so the replicated solve sees bit-identical systems (what the ring gives, too).
// Mailbox of a rank (hipMalloc, exported with hipIpcGetMemHandle, opened by the peers): slots [2][world][cap] doubles, flags [2][world] u32.  Two parities:
// a peer can be at most one call ahead (its call k+2 needs this rank's push k+1, which is stream-ordered behind this rank's reduce k), so the slot set being
// read in call k is never the one written by a peer's call k+1.  The push kernel never waits; the reduce kernel waits only for remote pushes -- no
// co-residency requirement, no deadlock by construction; the wait is bounded and poisons the result (NaN) + counts a timeout if a peer never arrives.
constexpr int P2P_MAX_WORLD = 16;
struct P2PDev {
  unsigned long long *box[P2P_MAX_WORLD];     // mailboxes, [rank] = this rank's own
  int world, rank;
  size_t cap;                                 // doubles per slot
};
struct P2PState {
  P2PDev dev{};
  void *mine = nullptr;
  void *peer[P2P_MAX_WORLD] = {};
  unsigned *d_done = nullptr;                 // push kernel: workgroups finished (last one raises the flags)
  unsigned *d_timeouts = nullptr;
  unsigned epoch = 0;
  size_t bytes = 0;
  bool connected = false;
  int mem_kind = 0;                           // 1: fine-grained device memory, 2: uncached, 3: plain hipMalloc (coarse-grained: last resort)
};

struct svs_comm {
  svs_ctx *ctx = nullptr;
  ncclComm_t comm = nullptr;
  P2PState *p2p = nullptr;                    // non-null: the one-shot transport instead of RCCL
  int rank = 0, world = 1;
  size_t n_calls = 0, n_doubles = 0;        // statistics (tests / bench): collectives issued, doubles reduced
};

namespace {
struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};
RcclApi *rccl_api(std::string *why) {
  static RcclApi api;
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names) if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);      // an instance the process already holds
    for (const char *n : names) if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!api.lib) api.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!api.lib) api.err = std::string("librccl not loadable: ") + (dlerror() ? dlerror() : "?");
    else {
      api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.lib, "ncclGetUniqueId"));
      api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.lib, "ncclCommInitRank"));
      api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
      api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.lib, "ncclAllReduce"));
      api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
      if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce) { api.err = "librccl lacks the nccl* entry points"; api.lib = nullptr; }
    }
  }
  if (!api.lib) { if (why) *why = api.err; return nullptr; }
  return &api;
}
}  // namespace

#define SVS_RCCL(ctx, api, call)                                                                      \
  do {                                                                                                \
    ncclResult_t r_ = (call);                                                                         \
    if (r_ != ncclSuccess) {                                                                          \
      (ctx)->err = std::string(#call " -> ") + ((api)->GetErrorString ? (api)->GetErrorString(r_) : "rccl error"); \
      return SVS_ERR_HIP;                                                                             \
    }                                                                                                 \
  } while (0)

extern "C" int svs_comm_get_unique_id(svs_ctx *ctx, svs_unique_id *out) {
  SVS_REQUIRE(ctx, ctx && out);
  static_assert(sizeof(svs_unique_id) == sizeof(ncclUniqueId), "svs_unique_id must mirror ncclUniqueId");
  RcclApi *api = rccl_api(&ctx->err);
  if (!api) return SVS_ERR_UNSUPPORTED;
  ncclUniqueId id;
  SVS_RCCL(ctx, api, api->GetUniqueId(&id));
  __builtin_memcpy(out->bytes, &id, sizeof id);
  return SVS_OK;
}

extern "C" int svs_comm_create(svs_ctx *ctx, const svs_unique_id *id, int rank, int world, svs_comm **out) {
  SVS_REQUIRE(ctx, ctx && id && out && world >= 1 && rank >= 0 && rank < world);
  RcclApi *api = rccl_api(&ctx->err);
  if (!api) return SVS_ERR_UNSUPPORTED;
  SVS_DEVICE(ctx);
  ncclUniqueId nid;
  __builtin_memcpy(&nid, id->bytes, sizeof nid);
  svs_comm *c = new svs_comm();
  c->ctx = ctx; c->rank = rank; c->world = world;
  ncclResult_t r = api->CommInitRank(&c->comm, world, nid, rank);
  if (r != ncclSuccess) { ctx->err = std::string("ncclCommInitRank -> ") + (api->GetErrorString ? api->GetErrorString(r) : "rccl error"); delete c; return SVS_ERR_HIP; }
  *out = c;
  return SVS_OK;
}

namespace {
__device__ __forceinline__ unsigned long long *p2p_slot(const P2PDev &D, int box, int parity, int from) {
  return D.box[box] + ((size_t)parity * D.world + from) * D.cap;
}
__device__ __forceinline__ unsigned *p2p_flags(const P2PDev &D, int box) { return reinterpret_cast<unsigned *>(D.box[box] + 2 * (size_t)D.world * D.cap); }

// write src[0..count) into slot [rank] of every peer; the workgroup that finishes last raises flag [rank] = epoch in every peer's mailbox
__global__ __launch_bounds__(256) void p2p_push_kernel(P2PDev D, const unsigned long long *__restrict__ src, size_t count, int parity, unsigned epoch,
                                                       unsigned *__restrict__ done) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
    const unsigned long long v = src[i];
    for (int q = 0; q < D.world; ++q)
      if (q != D.rank) __hip_atomic_store(p2p_slot(D, q, parity, D.rank) + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();                         // this thread's remote writes are out before its workgroup counts as done
  __syncthreads();
  __shared__ unsigned s_last;
  if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (s_last) {
    if (threadIdx.x == 0) __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // next launch on this stream starts from 0
    if ((int)threadIdx.x < D.world && (int)threadIdx.x != D.rank)
      __hip_atomic_store(p2p_flags(D, threadIdx.x) + parity * D.world + D.rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// wait for the peers' flags in the own mailbox, then buf[i] = sum over ranks 0..world-1 (own contribution from buf itself)
__global__ __launch_bounds__(256) void p2p_reduce_kernel(P2PDev D, double *__restrict__ buf, size_t count, int parity, unsigned epoch,
                                                         unsigned *__restrict__ timeouts) {
  __shared__ int s_bad;
  if (threadIdx.x == 0) s_bad = 0;
  __syncthreads();
  if ((int)threadIdx.x < D.world && (int)threadIdx.x != D.rank) {
    const unsigned *flag = p2p_flags(D, D.rank) + parity * D.world + threadIdx.x;
    bool ok = false;
    for (int spin = 0; spin < (1 << 24) && !ok; ++spin) {
      ok = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == epoch;
      if (!ok) __builtin_amdgcn_s_sleep(8);
    }
    if (!ok) { s_bad = 1; if (blockIdx.x == 0) atomicAdd(timeouts, 1u); }
  }
  __syncthreads();
  const bool bad = s_bad != 0;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
    double acc = 0.0;
    for (int q = 0; q < D.world; ++q) {
      const double v = q == D.rank ? buf[i]
                                   : __longlong_as_double((long long)__hip_atomic_load(p2p_slot(D, D.rank, parity, q) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
      acc = q == 0 ? v : acc + v;
    }
    buf[i] = bad ? __longlong_as_double(0x7ff8000000000000ll) : acc;      // a peer never arrived: the result must not look like a sum
  }
}
int p2p_allreduce(svs_comm *c, double *d_buf, size_t count) {
  svs_ctx *ctx = c->ctx;
  P2PState *S = c->p2p;
  SVS_REQUIRE(ctx, S->connected);
  SVS_DEVICE(ctx);
  for (size_t off = 0; off < count; off += S->dev.cap) {               // messages longer than a slot travel in slot-sized pieces, one epoch each
    const size_t n = std::min(S->dev.cap, count - off);
    const unsigned epoch = ++S->epoch;
    const int parity = (int)(epoch & 1u);
    const int grid = (int)std::min<size_t>((n + 1023) / 1024, 64);     // a small grid: the exchange is latency-bound, and the device stays free for the peers' pushes when ranks share a GPU (tests)
    hipLaunchKernelGGL(p2p_push_kernel, dim3(grid), dim3(256), 0, ctx->stream, S->dev, reinterpret_cast<const unsigned long long *>(d_buf + off), n, parity, epoch,
                       S->d_done);
    SVS_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(p2p_reduce_kernel, dim3(grid), dim3(256), 0, ctx->stream, S->dev, d_buf + off, n, parity, epoch, S->d_timeouts);
    SVS_LAUNCH_CHECK(ctx);
  }
  return SVS_OK;
}
void p2p_free(P2PState *S) {
  if (!S) return;
  for (int q = 0; q < P2P_MAX_WORLD; ++q) if (S->peer[q] && S->peer[q] != S->mine) (void)hipIpcCloseMemHandle(S->peer[q]);
  if (S->mine) (void)hipFree(S->mine);
  if (S->d_done) (void)hipFree(S->d_done);
  delete S;
}
}  // namespace

/* one-shot transport, step 1 (every rank): allocate the mailbox on the context's device and export it */
extern "C" int svs_comm_create_p2p(svs_ctx *ctx, int rank, int world, size_t capacity_doubles, svs_comm **out, svs_ipc_handle *h_mine) {
  SVS_REQUIRE(ctx, ctx && out && h_mine && world >= 1 && world <= P2P_MAX_WORLD && rank >= 0 && rank < world && capacity_doubles >= 32);
  static_assert(sizeof(svs_ipc_handle) == sizeof(hipIpcMemHandle_t), "svs_ipc_handle must mirror hipIpcMemHandle_t");
  SVS_DEVICE(ctx);
  svs_comm *c = new svs_comm();
  c->ctx = ctx; c->rank = rank; c->world = world;
  P2PState *S = c->p2p = new P2PState();
  auto fail = [&](int rc) { p2p_free(S); delete c; return rc; };
  S->dev.world = world; S->dev.rank = rank; S->dev.cap = capacity_doubles;
  S->bytes = 2 * (size_t)world * capacity_doubles * sizeof(double) + 2 * (size_t)world * sizeof(unsigned);
  // The mailbox is written by PEER devices while this device's reduce kernel polls it.  Plain hipMalloc memory is coarse-grained: coherence with other agents is
  // only guaranteed at kernel boundaries (a per-XCD L2 may keep serving a stale line of a flag or slot).  RCCL allocates its flags fine-grained / uncached for
  // that reason, and so does this: fine-grained first, uncached second, and plain memory only if neither can be allocated AND exported (reported by
  // svs_comm_transport, so that a multi-GPU run states what it ran on).  The kernels' sc1 / system-scope accesses stay as they are.
  __builtin_memset(h_mine, 0, sizeof *h_mine);
  const unsigned kinds[3] = {hipDeviceMallocFinegrained, hipDeviceMallocUncached, hipDeviceMallocDefault};
  for (int k = 0; k < 3 && !S->mine; ++k) {
    void *p = nullptr;
    if ((k < 2 ? hipExtMallocWithFlags(&p, S->bytes, kinds[k]) : hipMalloc(&p, S->bytes)) != hipSuccess || !p) { (void)hipGetLastError(); continue; }
    if (world > 1) {
      hipIpcMemHandle_t h;
      if (hipIpcGetMemHandle(&h, p) != hipSuccess) {
        (void)hipGetLastError(); (void)hipFree(p);
        if (k == 2) { ctx->err = "hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set on this driver)"; return fail(SVS_ERR_HIP); }
        continue;
      }
      __builtin_memcpy(h_mine->bytes, &h, sizeof h);
    }
    S->mine = p; S->mem_kind = k + 1;
  }
  if (!S->mine || hipMalloc((void **)&S->d_done, 2 * sizeof(unsigned)) != hipSuccess) { ctx->err = "p2p mailbox allocation failed"; return fail(SVS_ERR_HIP); }
  S->d_timeouts = S->d_done + 1;
  if (hipMemset(S->mine, 0, S->bytes) != hipSuccess || hipMemset(S->d_done, 0, 2 * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    ctx->err = "p2p mailbox initialisation failed"; return fail(SVS_ERR_HIP);
  }
  S->peer[rank] = S->mine;
  S->dev.box[rank] = static_cast<unsigned long long *>(S->mine);
  if (world == 1) S->connected = true;
  *out = c;
  return SVS_OK;
}
/* step 2, after the handles of all ranks have been gathered out of band (h_all [world], own entry ignored): map the peers' mailboxes */
extern "C" int svs_comm_connect_p2p(svs_comm *c, const svs_ipc_handle *h_all) {
  svs_ctx *ctx = c ? c->ctx : nullptr;
  SVS_REQUIRE(ctx, c && c->p2p && h_all && !c->p2p->connected);
  SVS_DEVICE(ctx);
  P2PState *S = c->p2p;
  for (int q = 0; q < c->world; ++q) {
    if (q == c->rank) continue;
    hipIpcMemHandle_t h;
    __builtin_memcpy(&h, h_all[q].bytes, sizeof h);
    if (hipIpcOpenMemHandle(&S->peer[q], h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
      (void)hipGetLastError();
      ctx->err = "hipIpcOpenMemHandle failed for rank " + std::to_string(q) + " (no peer access between the two devices?)";
      return SVS_ERR_HIP;
    }
    S->dev.box[q] = static_cast<unsigned long long *>(S->peer[q]);
  }
  S->connected = true;
  return SVS_OK;
}
/* 0: RCCL; one-shot P2P: 1 (mailbox in fine-grained device memory), 2 (uncached), 3 (plain hipMalloc); *timeouts = reduce kernels that gave up waiting for a
   peer (their output is NaN) -- blocking */
extern "C" int svs_comm_transport(svs_comm *c, int32_t *kind, uint32_t *timeouts) {
  if (!c) return SVS_ERR_INVALID;
  if (kind) *kind = c->p2p ? c->p2p->mem_kind : 0;
  if (timeouts) {
    *timeouts = 0;
    if (c->p2p) {
      svs_ctx *ctx = c->ctx;
      SVS_DEVICE(ctx);
      SVS_HIP(ctx, hipMemcpyAsync(timeouts, c->p2p->d_timeouts, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
      SVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
  }
  return SVS_OK;
}

extern "C" int svs_comm_destroy(svs_comm *c) {
  if (!c) return SVS_OK;
  RcclApi *api = c->p2p ? nullptr : rccl_api(nullptr);
  (void)hipStreamSynchronize(c->ctx->stream);
  if (api && c->comm) (void)api->CommDestroy(c->comm);
  p2p_free(c->p2p);
  delete c;
  return SVS_OK;
}

extern "C" int svs_comm_allreduce_f64(svs_comm *c, void *d_buf, size_t count) {
  svs_ctx *ctx = c ? c->ctx : nullptr;
  SVS_REQUIRE(ctx, c && d_buf);
  if (count == 0) return SVS_OK;
  if (c->p2p) {
    if (c->world > 1) { const int rc = p2p_allreduce(c, static_cast<double *>(d_buf), count); if (rc) return rc; }
    ++c->n_calls; c->n_doubles += count;
    return SVS_OK;
  }
  RcclApi *api = rccl_api(&ctx->err);
  if (!api) return SVS_ERR_UNSUPPORTED;
  SVS_RCCL(ctx, api, api->AllReduce(d_buf, d_buf, count, ncclDouble, ncclSum, c->comm, ctx->stream));
  ++c->n_calls; c->n_doubles += count;
  return SVS_OK;
}

extern "C" int svs_comm_stats(svs_comm *c, int32_t *rank, int32_t *world, uint64_t *n_calls, uint64_t *n_doubles) {
  if (!c) return SVS_ERR_INVALID;
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (n_calls) *n_calls = c->n_calls;
  if (n_doubles) *n_doubles = c->n_doubles;
  return SVS_OK;
}

// hook form of the same collective (svs_allreduce_fn), used by svs_ba when a communicator is attached
int svs_comm_allreduce_hook(void *d_buf, size_t count, void *user) { return svs_comm_allreduce_f64(static_cast<svs_comm *>(user), d_buf, count); }
