// comm.hip -- the multi-GPU exchange of the detector path as C entry points: one process per GPU, replicated engines, and ONE
// collective -- the all-gather of fixed-size detection slots (include/unina_mi355.h "multi-GPU"). The Python
// bench drives the same exchange through torch.distributed (gather.py); a C / C++ consumer (tools/node_harness.cpp, a ROS 2
// node with several GPUs) has no torch, so the library offers it over RCCL directly. RCCL is loaded on FIRST USE (dlopen):
// single-GPU consumers -- every reference deployment, perception_node.cpp:472,802 -- never pay for it and the library has no
// link-time dependency on it. Host code only.
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include <hip/hip_runtime.h>

#include "../../include/unina_mi355.h"

namespace {

// the handful of RCCL (= NCCL API) symbols used, with the library's own types spelled out (rccl.h: ncclComm_t is an opaque
// pointer, ncclResult_t an enum with ncclSuccess = 0, ncclUniqueId 128 opaque bytes, ncclUint8 = 1)
struct RcclId { char internal[UNINA_COMM_ID_BYTES]; };
typedef int (*GetUniqueIdFn)(RcclId*);
typedef int (*CommInitRankFn)(void**, int, RcclId, int);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*CommDestroyFn)(void*);
typedef const char* (*GetErrorStringFn)(int);
constexpr int kRcclUint8 = 1;

struct Rccl {
  void* so = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  AllGatherFn all_gather = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  GetErrorStringFn error_string = nullptr;
  std::string error;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.so) break;
    }
    if (!r.so) {
      r.error = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?");
      return;
    }
    r.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(r.so, "ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(r.so, "ncclCommInitRank"));
    r.all_gather = reinterpret_cast<AllGatherFn>(dlsym(r.so, "ncclAllGather"));
    r.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(r.so, "ncclCommDestroy"));
    r.error_string = reinterpret_cast<GetErrorStringFn>(dlsym(r.so, "ncclGetErrorString"));
    if (!r.get_unique_id || !r.comm_init_rank || !r.all_gather || !r.comm_destroy) r.error = "librccl lacks an expected symbol";
  });
  return r;
}

thread_local std::string g_comm_error;
int fail(int code, const std::string& msg) {
  g_comm_error = msg;
  return code;
}
std::string rccl_msg(const char* call, int rc) {
  Rccl& r = rccl();
  return std::string(call) + ": " + (r.error_string ? r.error_string(rc) : "RCCL error") + " (" + std::to_string(rc) + ")";
}

}  // namespace

struct unina_comm {
  void* comm = nullptr;
  int rank = 0, world = 1, device = 0;
};

extern "C" {

const char* unina_comm_last_error(void) { return g_comm_error.c_str(); }

int unina_comm_unique_id(void* id128) {
  if (!id128) return fail(UNINA_ERR_ARG, "unina_comm_unique_id: null id buffer");
  Rccl& r = rccl();
  if (!r.error.empty()) return fail(UNINA_ERR_UNSUPPORTED, r.error);
  RcclId id;
  const int rc = r.get_unique_id(&id);
  if (rc) return fail(UNINA_ERR_HIP, rccl_msg("ncclGetUniqueId", rc));
  memcpy(id128, &id, sizeof id);
  return UNINA_OK;
}

int unina_comm_init(unina_comm** out, const void* id128, int rank, int world, int device_id) {
  if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return fail(UNINA_ERR_ARG, "unina_comm_init: bad arguments");
  *out = nullptr;
  Rccl& r = rccl();
  if (!r.error.empty()) return fail(UNINA_ERR_UNSUPPORTED, r.error);
  hipError_t he = hipSetDevice(device_id);
  if (he != hipSuccess) return fail(UNINA_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(he));
  RcclId id;
  memcpy(&id, id128, sizeof id);
  unina_comm* c = new unina_comm;
  c->rank = rank; c->world = world; c->device = device_id;
  const int rc = r.comm_init_rank(&c->comm, world, id, rank);
  if (rc) {
    delete c;
    return fail(UNINA_ERR_HIP, rccl_msg("ncclCommInitRank", rc));
  }
  *out = c;
  return UNINA_OK;
}

int unina_comm_all_gather(unina_comm* c, const void* d_send, void* d_recv, size_t bytes_per_rank, hipStream_t stream) {
  if (!c || !c->comm || !d_send || !d_recv || bytes_per_rank == 0) return fail(UNINA_ERR_ARG, "unina_comm_all_gather: bad arguments");
  const int rc = rccl().all_gather(d_send, d_recv, bytes_per_rank, kRcclUint8, c->comm, stream);
  if (rc) return fail(UNINA_ERR_HIP, rccl_msg("ncclAllGather", rc));
  return UNINA_OK;
}

int unina_comm_rank(const unina_comm* c) { return c ? c->rank : -1; }
int unina_comm_world(const unina_comm* c) { return c ? c->world : -1; }

void unina_comm_destroy(unina_comm* c) {
  if (!c) return;
  if (c->comm) rccl().comm_destroy(c->comm);
  delete c;
}

}  // extern "C"
