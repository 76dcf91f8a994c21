// spgan_allreduce_flat: the data-parallel exchange of the train step (replaces nn.DataParallel's per-call
// scatter / broadcast / gather, Generation/model.py:79-84) as a native entry point -- one RCCL all-reduce (sum) of a network's flat
// gradient buffer, in place, on the caller's stream.  Because it is enqueued on the caller's stream and RCCL supports stream
// capture, a captured train step can contain its two collectives instead of being cut into graphs around them.
//
// RCCL is reached through dlopen at first use: libspgan_hip.so keeps no link-time dependency on librccl (single-GPU users never
// load it), and the host language needs no RCCL binding.  Rendezvous stays with the host: rank 0 obtains the 128-byte unique id
// (spgan_comm_unique_id) and hands it to the other ranks by whatever channel the host program has (torch.distributed's store / a
// broadcast; MPI; a file), then every rank calls spgan_comm_init.  Errors come back as status codes (SPGAN_ECOMM = RCCL missing or
// an RCCL call failed; spgan_comm_last_error(comm) gives the RCCL code of that communicator's last failure); nothing here synchronises the device.
#include <dlfcn.h>
#include <string.h>

#include "common.hpp"

namespace {

typedef struct { char internal[128]; } UniqueId;   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;                                // ncclComm_t
constexpr int kFloat32 = 7, kSum = 0;              // ncclFloat32, ncclSum (rccl.h)

struct Api {
  void* lib = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*CommCount)(Comm, int*) = nullptr;
  bool ok = false;
};

// No process-global mutable state: the RCCL code of a failed call is kept IN the communicator it failed on; failures
// before a communicator exists (unique id, init) are kept per calling thread.
struct Handle {
  Comm c = nullptr;
  int last_error = 0;
};
thread_local int t_last_error = 0;

Api& api() {
  static Api a = [] {
    Api x;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      x.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (x.lib) break;
    }
    if (!x.lib) return x;
    x.GetUniqueId = reinterpret_cast<int (*)(UniqueId*)>(dlsym(x.lib, "ncclGetUniqueId"));
    x.CommInitRank = reinterpret_cast<int (*)(Comm*, int, UniqueId, int)>(dlsym(x.lib, "ncclCommInitRank"));
    x.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, Comm, hipStream_t)>(dlsym(x.lib, "ncclAllReduce"));
    x.CommDestroy = reinterpret_cast<int (*)(Comm)>(dlsym(x.lib, "ncclCommDestroy"));
    x.CommCount = reinterpret_cast<int (*)(Comm, int*)>(dlsym(x.lib, "ncclCommCount"));
    x.ok = x.GetUniqueId && x.CommInitRank && x.AllReduce && x.CommDestroy;
    return x;
  }();
  return a;
}

inline int rc(int nccl_result, Handle* h = nullptr) {
  if (nccl_result == 0) return SPGAN_OK;
  if (h) h->last_error = nccl_result;
  else t_last_error = nccl_result;
  return SPGAN_ECOMM;
}

}  // namespace

extern "C" int spgan_comm_available(void) { return api().ok ? 1 : 0; }
extern "C" int spgan_comm_last_error(void* comm) { return comm ? static_cast<Handle*>(comm)->last_error : t_last_error; }

extern "C" int spgan_comm_unique_id(void* id128) {
  SPGAN_CHECK_ARG(id128);
  if (!api().ok) return SPGAN_ECOMM;
  UniqueId id;
  const int r = rc(api().GetUniqueId(&id));
  if (r == SPGAN_OK) memcpy(id128, id.internal, sizeof(id.internal));
  return r;
}

extern "C" int spgan_comm_init(const void* id128, int rank, int world, void** comm) {
  SPGAN_CHECK_ARG(id128 && comm && world >= 1 && rank >= 0 && rank < world);
  if (!api().ok) return SPGAN_ECOMM;
  UniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  Comm c = nullptr;
  const int r = rc(api().CommInitRank(&c, world, id, rank));   // binds to the calling thread's current HIP device
  if (r == SPGAN_OK) {
    Handle* h = new Handle;
    h->c = c;
    *comm = h;
  }
  return r;
}

extern "C" int spgan_comm_world(void* comm) {
  if (!comm || !api().ok || !api().CommCount) return -1;
  int n = -1;
  return api().CommCount(static_cast<Handle*>(comm)->c, &n) == 0 ? n : -1;
}

extern "C" int spgan_allreduce_flat(void* comm, float* buf, size_t n, spgan_stream_t s) {
  SPGAN_CHECK_ARG(comm && buf);
  if (n == 0) return SPGAN_OK;
  if (!api().ok) return SPGAN_ECOMM;
  Handle* h = static_cast<Handle*>(comm);
  return rc(api().AllReduce(buf, buf, n, kFloat32, kSum, h->c, (hipStream_t)s), h);   // in place, sum; the caller scales (Adam's grad_scale)
}

extern "C" int spgan_comm_destroy(void* comm) {
  if (!comm) return SPGAN_OK;
  if (!api().ok) return SPGAN_ECOMM;
  Handle* h = static_cast<Handle*>(comm);
  const int r = rc(api().CommDestroy(h->c));
  delete h;
  return r;
}
