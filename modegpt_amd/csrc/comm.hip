// The one data-path collective of the layer-sharded run behind the C ABI: an all-gather of the packed per-layer records over
// RCCL (xGMI inside a node).  The engine's own driver goes through torch.distributed (backend "nccl" = RCCL); these entry
// points give a host WITHOUT torch -- a maintainer of the reference binding the library through ctypes -- the same step.
// RCCL is resolved at run time (dlopen of librccl.so.1 on first use): the library has no link-time dependency on it, and a
// process that already holds RCCL (PyTorch) shares that instance.
#include <dlfcn.h>

#include "common.hpp"

namespace mdg {
namespace {

constexpr int UNIQUE_ID_BYTES = 128;   // NCCL_UNIQUE_ID_BYTES
struct UniqueId { char internal[UNIQUE_ID_BYTES]; };
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*CommDestroyFn)(void*);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);

struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  AllGatherFn all_gather = nullptr;
  GetErrorStringFn error_string = nullptr;
};

const Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) break;
    }
    if (r.handle) {
      r.get_unique_id = (GetUniqueIdFn)dlsym(r.handle, "ncclGetUniqueId");
      r.comm_init_rank = (CommInitRankFn)dlsym(r.handle, "ncclCommInitRank");
      r.comm_destroy = (CommDestroyFn)dlsym(r.handle, "ncclCommDestroy");
      r.all_gather = (AllGatherFn)dlsym(r.handle, "ncclAllGather");
      r.error_string = (GetErrorStringFn)dlsym(r.handle, "ncclGetErrorString");
    }
  }
  return (r.handle && r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_gather) ? &r : nullptr;
}

int fail(const Rccl* r, const char* what, int code) {
  set_error("%s failed: %s (RCCL result %d)", what, (r && r->error_string) ? r->error_string(code) : "?", code);
  return MDG_ERR_HIP;
}

}  // namespace
}  // namespace mdg

using namespace mdg;

#define MDG_NEED_RCCL(r)                                                                     \
  const Rccl* r = rccl();                                                                    \
  if (!r) {                                                                                  \
    set_error("RCCL not available: dlopen(librccl.so.1) failed (%s)", dlerror() ? dlerror() : "symbols missing"); \
    return MDG_ERR_HIP;                                                                      \
  }

extern "C" int mdg_comm_unique_id(void* id128) {
  MDG_CLEAR();
  MDG_CHECK_ARG(id128, "mdg_comm_unique_id: null pointer");
  MDG_NEED_RCCL(r);
  UniqueId id;
  const int rc = r->get_unique_id(&id);
  if (rc != 0) return fail(r, "ncclGetUniqueId", rc);
  memcpy(id128, id.internal, UNIQUE_ID_BYTES);
  return MDG_OK;
}

extern "C" int mdg_comm_init(void** comm, int world, int rank, const void* id128) {
  MDG_CLEAR();
  MDG_CHECK_ARG(comm && id128 && world >= 1 && rank >= 0 && rank < world, "mdg_comm_init: bad arguments (world=%d rank=%d)", world, rank);
  MDG_NEED_RCCL(r);
  UniqueId id;
  memcpy(id.internal, id128, UNIQUE_ID_BYTES);
  const int rc = r->comm_init_rank(comm, world, id, rank);
  if (rc != 0) return fail(r, "ncclCommInitRank", rc);
  return MDG_OK;
}

extern "C" int mdg_comm_destroy(void* comm) {
  MDG_CLEAR();
  if (!comm) return MDG_OK;
  MDG_NEED_RCCL(r);
  const int rc = r->comm_destroy(comm);
  if (rc != 0) return fail(r, "ncclCommDestroy", rc);
  return MDG_OK;
}

extern "C" int mdg_allgather_layers(const void* send, void* recv, size_t bytes_per_rank, void* comm, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(send && recv && comm && bytes_per_rank > 0, "mdg_allgather_layers: bad arguments");
  MDG_NEED_RCCL(r);
  const int rc = r->all_gather(send, recv, bytes_per_rank, /* ncclUint8 */ 1, comm, (hipStream_t)stream);
  if (rc != 0) return fail(r, "ncclAllGather", rc);
  return MDG_OK;
}
