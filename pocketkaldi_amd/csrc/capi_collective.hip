// capi_collective.hip -- the one collective of the path: the weight-blob broadcast over the caller's RCCL
// communicator (utterance-sharded ranks each hold a full replica).
#include <hip/hip_runtime.h>
#include <ctype.h>
#include <dlfcn.h>
#include <math.h>
#include <cmath>

#include <algorithm>
#include <string>
#include <mutex>
#include <utility>
#include <unordered_set>
#include <vector>

#include "pk_host.h"

using namespace pkmi;
using namespace pkhost;

// ---- the one collective of the path: weight-blob broadcast over the caller's RCCL communicator.
// RCCL's C API, bound at run time (rccl.h: ncclBroadcast, ncclGetErrorString; ncclUint8 = 1).
namespace {
typedef int (*NcclBroadcastFn)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef const char *(*NcclErrorStringFn)(int);
std::mutex g_rccl_mu;
NcclBroadcastFn g_nccl_broadcast = nullptr;
NcclErrorStringFn g_nccl_error_string = nullptr;

int BindRccl() {
  std::lock_guard<std::mutex> g(g_rccl_mu);
  if (g_nccl_broadcast) return 0;
  // The communicator belongs to ONE copy of RCCL: the one the caller created it with, which is
  // therefore already loaded.  Bind to that copy and never load a second one behind the caller's
  // back (a foreign ncclComm_t handed to another copy is undefined behaviour):
  //   1. $PK_MI355_RCCL_LIB, when set, names the library explicitly (loaded if need be);
  //   2. a copy visible in the global scope (a host linked with -lrccl);
  //   3. a copy loaded RTLD_LOCAL (Python / torch): found by soname with RTLD_NOLOAD.
  void *h = nullptr;
  void *sym = nullptr;
  const char *path = getenv("PK_MI355_RCCL_LIB");
  if (path && *path) {
    h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return Fail(PK_MI355_E_DEVICE, "PK_MI355_RCCL_LIB=%s cannot be opened: %s", path, dlerror());
    sym = dlsym(h, "ncclBroadcast");
  } else {
    sym = dlsym(RTLD_DEFAULT, "ncclBroadcast");
    if (!sym) {
      h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
      if (!h)
        return Fail(PK_MI355_E_DEVICE, "no RCCL is loaded in this process (the communicator's library must be; "
                                       "or name it in PK_MI355_RCCL_LIB)");
      sym = dlsym(h, "ncclBroadcast");
    }
  }
  if (!sym) return Fail(PK_MI355_E_DEVICE, "ncclBroadcast not found in RCCL");
  void *es = h ? dlsym(h, "ncclGetErrorString") : dlsym(RTLD_DEFAULT, "ncclGetErrorString");
  g_nccl_error_string = reinterpret_cast<NcclErrorStringFn>(es);
  g_nccl_broadcast = reinterpret_cast<NcclBroadcastFn>(sym);
  return 0;
}
}  // namespace

extern "C" {

int pk_mi355_am_broadcast_from(pk_mi355_am_t *am, pk_mi355_am_t *src, void *rccl_comm, int root, void *stream) {
  if (!am || !am->finalized) return Fail(PK_MI355_E_STATE, "model not finalized");
  if (src && (!src->finalized || src->blob_floats != am->blob_floats || src->device != am->device))
    return Fail(PK_MI355_E_INVALID, "source model does not have the destination's blob layout / device");
  if (!rccl_comm) return Fail(PK_MI355_E_INVALID, "null RCCL communicator");
  if (root < 0) return Fail(PK_MI355_E_INVALID, "bad root rank %d", root);
  int rc = UseDevice(am->device);
  if (rc || (rc = BindRccl())) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipStream_t own = nullptr;
  if (!s) {
    HIP_TRY(hipStreamCreateWithFlags(&own, hipStreamNonBlocking));
    s = own;
  }
  // receive buffer = this rank's blob (same size on every rank: the layout depends on the layer
  // structure only); send buffer = the same blob (in place) or, on the root, `src`'s
  const int nr = g_nccl_broadcast(src ? src->d_blob : am->d_blob, am->d_blob, am->blob_floats * sizeof(float),
                                  /*ncclUint8*/ 1, root, rccl_comm, s);
  if (nr != 0) {
    if (own) hipStreamDestroy(own);
    return Fail(PK_MI355_E_DEVICE, "ncclBroadcast failed: %s", g_nccl_error_string ? g_nccl_error_string(nr) : "?");
  }
  am->exps_stale = IsF16(am->precision);     // the blob's exponent words now are the root's
  if (own) {
    hipError_t e = hipStreamSynchronize(own);
    hipStreamDestroy(own);
    if (e != hipSuccess) return Fail(PK_MI355_E_DEVICE, "broadcast stream: %s", hipGetErrorString(e));
    return RefreshExps(am);
  }
  return 0;
}

int pk_mi355_am_broadcast(pk_mi355_am_t *am, void *rccl_comm, int root, void *stream) {
  return pk_mi355_am_broadcast_from(am, nullptr, rccl_comm, root, stream);
}

}  // extern "C"
