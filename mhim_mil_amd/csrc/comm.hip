// comm.hip — the communicator handle of the C-ABI: mhimx_comm_{unique_id, init, allreduce, destroy}.
//
// The data-parallel update of the reference is DDP's bucketed all-reduce (engines/base_engine.py via options.py:287); here it is ONE
// collective on the flat gradient buffer.  This file puts that collective behind the C boundary so that a host in any
// language can drive it without torch.distributed: RCCL is loaded at run time (dlopen: libmhimx.so has no link-time dependency on it
// and loads on a machine without RCCL; the product's default path through torch.distributed uses the same library).
//   mode 0: ncclAllReduce (RCCL picks ring / tree);
//   mode 1: reduce-scatter + all-gather on count / world slices - on the xGMI full mesh every rank then talks to its 7 peers at once
//           with 1/8 of the buffer per link instead of walking a ring; needs count % world == 0, else falls back to mode 0.
// Enqueue-only on the caller's stream like every other entry point; the sum is in place.
#include <dlfcn.h>
#include <string.h>
#include <mutex>

#include "common.hpp"

namespace mhimx {
namespace {

struct NcclId { char internal[128]; };
typedef int (*fn_get_id)(NcclId*);
typedef int (*fn_init_rank)(void**, int, NcclId, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_reduce_scatter)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_allgather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_errstr)(int);

struct Rccl {
  void* h = nullptr;
  fn_get_id get_id = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_allreduce allreduce = nullptr;
  fn_reduce_scatter reduce_scatter = nullptr;
  fn_allgather allgather = nullptr;
  fn_destroy destroy = nullptr;
  fn_errstr errstr = nullptr;
  std::string err;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (r.h) break;
    }
    if (!r.h) {
      const char* e = dlerror();               // (dlerror() clears the message: read it once)
      r.err = std::string("dlopen(librccl.so): ") + (e ? e : "not found");
      return;
    }
    r.get_id = (fn_get_id)dlsym(r.h, "ncclGetUniqueId");
    r.init_rank = (fn_init_rank)dlsym(r.h, "ncclCommInitRank");
    r.allreduce = (fn_allreduce)dlsym(r.h, "ncclAllReduce");
    r.reduce_scatter = (fn_reduce_scatter)dlsym(r.h, "ncclReduceScatter");
    r.allgather = (fn_allgather)dlsym(r.h, "ncclAllGather");
    r.destroy = (fn_destroy)dlsym(r.h, "ncclCommDestroy");
    r.errstr = (fn_errstr)dlsym(r.h, "ncclGetErrorString");
    if (!r.get_id || !r.init_rank || !r.allreduce || !r.reduce_scatter || !r.allgather || !r.destroy) r.err = "librccl.so lacks an nccl* symbol";
  });
  return r;
}

constexpr int NCCL_FLOAT = 7, NCCL_SUM = 0;

int nccl_fail(const char* what, int code) {
  Rccl& r = rccl();
  return fail(-2, "%s: RCCL error %d (%s)", what, code, r.errstr ? r.errstr(code) : "?");
}

}  // namespace
}  // namespace mhimx

struct mhimx_comm {
  void* nccl;
  int rank, world;
};

using namespace mhimx;

extern "C" int mhimx_comm_unique_id(void* id128) {
  MHIMX_CHECK_ARG(id128, "comm_unique_id: null buffer");
  Rccl& r = rccl();
  if (!r.err.empty()) return fail(-2, "comm_unique_id: %s", r.err.c_str());
  NcclId id;
  if (int e = r.get_id(&id)) return nccl_fail("ncclGetUniqueId", e);
  memcpy(id128, &id, sizeof(id));
  return 0;
}

extern "C" int mhimx_comm_init(mhimx_comm** out, const void* id128, int32_t rank, int32_t world) {
  MHIMX_CHECK_ARG(out && id128 && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments");
  Rccl& r = rccl();
  if (!r.err.empty()) return fail(-2, "comm_init: %s", r.err.c_str());
  NcclId id;
  memcpy(&id, id128, sizeof(id));
  void* c = nullptr;
  if (int e = r.init_rank(&c, world, id, rank)) return nccl_fail("ncclCommInitRank", e);
  *out = new mhimx_comm{c, rank, world};
  return 0;
}

extern "C" int mhimx_comm_allreduce(mhimx_comm* c, void* stream, float* buf, int64_t count, int32_t mode) {
  MHIMX_CHECK_ARG(c && c->nccl && buf && count >= 0, "comm_allreduce: bad arguments");
  if (count == 0) return 0;
  Rccl& r = rccl();
  hipStream_t st = (hipStream_t)stream;
  if (mode == 1 && c->world > 1 && count % c->world == 0) {
    const size_t slice = (size_t)(count / c->world);
    if (int e = r.reduce_scatter(buf, buf + slice * c->rank, slice, NCCL_FLOAT, NCCL_SUM, c->nccl, st)) return nccl_fail("ncclReduceScatter", e);
    if (int e = r.allgather(buf + slice * c->rank, buf, slice, NCCL_FLOAT, c->nccl, st)) return nccl_fail("ncclAllGather", e);
    return 0;
  }
  if (int e = r.allreduce(buf, buf, (size_t)count, NCCL_FLOAT, NCCL_SUM, c->nccl, st)) return nccl_fail("ncclAllReduce", e);
  return 0;
}

extern "C" int mhimx_comm_destroy(mhimx_comm* c) {
  if (!c) return 0;
  int e = 0;
  if (c->nccl) e = rccl().destroy(c->nccl);
  delete c;
  return e ? nccl_fail("ncclCommDestroy", e) : 0;
}
