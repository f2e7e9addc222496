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

struct svs_comm {
  svs_ctx *ctx = nullptr;
  ncclComm_t comm = nullptr;
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

extern "C" int svs_comm_destroy(svs_comm *c) {
  if (!c) return SVS_OK;
  RcclApi *api = rccl_api(nullptr);
  (void)hipStreamSynchronize(c->ctx->stream);
  if (api && c->comm) (void)api->CommDestroy(c->comm);
  delete c;
  return SVS_OK;
}

extern "C" int svs_comm_allreduce_f64(svs_comm *c, void *d_buf, size_t count) {
  svs_ctx *ctx = c ? c->ctx : nullptr;
  SVS_REQUIRE(ctx, c && d_buf);
  RcclApi *api = rccl_api(&ctx->err);
  if (!api) return SVS_ERR_UNSUPPORTED;
  if (count == 0) return SVS_OK;
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
