// Gradient collective behind the C ABI: a thin binding of RCCL (the xGMI collective library of ROCm) for hosts that
// do not run torch.distributed.  Replaces what DistributedDataParallel does for the reference
// (app/vjepa/train.py:295-297): sum-all-reduce of slices of the flat fp32 gradient arena on a communication stream.
//
// librccl is NOT a link-time dependency of libvjepa_hip.so: it is dlopen()ed on the first vj_comm_* call, preferring
// the copy already mapped into the process (PyTorch-ROCm ships and loads its own librccl.so), so a process never ends
// up with two RCCL runtimes.  One process per GPU; the communicator is created with ncclCommInitRank from a 128-byte
// unique id that rank 0 obtains from vj_comm_unique_id and distributes out of band (file, env, MPI, a TCP store).
#include "common.hpp"
#include "../../include/vjepa_hip.h"
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <dlfcn.h>
#include <link.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <rccl/rccl.h>

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl& rccl() {
  std::call_once(g_rccl_once, [] {
    const char* names[] = {"librccl.so", "librccl.so.1"};
    // a copy that is ALREADY mapped into the process wins (PyTorch-ROCm ships its own librccl.so and loads it by path: a lookup
    // by soname does not find it, and binding the system copy next to it would leave the process with two RCCL runtimes):
    // walk the loaded objects for a path that names librccl and re-open exactly that file
    struct Found { char path[1024]; } found = {{0}};
    dl_iterate_phdr(
        [](struct dl_phdr_info* info, size_t, void* data) -> int {
          // the BASENAME must be librccl.so[.N...]: plugin objects such as librccl-net.so also contain "librccl"
          if (!info->dlpi_name) return 0;
          const char* base = strrchr(info->dlpi_name, '/');
          base = base ? base + 1 : info->dlpi_name;
          if (strncmp(base, "librccl.so", 10) == 0 && (base[10] == 0 || base[10] == '.')) {
            strncpy(((Found*)data)->path, info->dlpi_name, sizeof(Found::path) - 1);
            return 1;
          }
          return 0;
        },
        &found);
    if (found.path[0]) {
      g_rccl.h = dlopen(found.path, RTLD_NOW | RTLD_NOLOAD);
      if (g_rccl.h && !dlsym(g_rccl.h, "ncclGetUniqueId")) {   // not an RCCL runtime after all: fall through to the names below
        dlclose(g_rccl.h);
        g_rccl.h = nullptr;
      }
    }
    for (const char* n : names) {
      if (g_rccl.h) break;
      g_rccl.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    }
    for (int i = 0; i < 2 && !g_rccl.h; i++) g_rccl.h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!g_rccl.h) g_rccl.h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!g_rccl.h) return;
#define VJ_SYM(field, name) g_rccl.field = (decltype(g_rccl.field))dlsym(g_rccl.h, name)
    VJ_SYM(GetUniqueId, "ncclGetUniqueId");
    VJ_SYM(CommInitRank, "ncclCommInitRank");
    VJ_SYM(CommDestroy, "ncclCommDestroy");
    VJ_SYM(AllReduce, "ncclAllReduce");
    VJ_SYM(Broadcast, "ncclBroadcast");
    VJ_SYM(GetErrorString, "ncclGetErrorString");
#undef VJ_SYM
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.Broadcast &&
                g_rccl.GetErrorString;
    if (getenv("VJ_COMM_DEBUG")) {   // which librccl did we bind?  (two RCCL runtimes in one process is the thing to avoid)
      Dl_info info;
      if (g_rccl.AllReduce && dladdr((void*)g_rccl.AllReduce, &info) && info.dli_fname)
        fprintf(stderr, "libvjepa_hip: vj_comm_* bound to %s\n", info.dli_fname);
    }
  });
  return g_rccl;
}

int fail(const char* what, ncclResult_t r) {
  vj_set_error("%s: RCCL error %d (%s)", what, (int)r, rccl().GetErrorString ? rccl().GetErrorString(r) : "?");
  return 1000 + (int)r;
}
}  // namespace

#define VJ_NEED_RCCL(what)                                                                             \
  do {                                                                                                 \
    if (!rccl().ok) {                                                                                  \
      vj_set_error("%s: librccl.so could not be loaded (is ROCm's RCCL installed?): %s", what, dlerror()); \
      return -3;                                                                                       \
    }                                                                                                  \
  } while (0)

extern "C" int64_t vj_comm_unique_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

extern "C" int vj_comm_unique_id(void* id_out) {
  VJ_NEED_RCCL("vj_comm_unique_id");
  VJ_CHECK_ARG(id_out != nullptr, "vj_comm_unique_id: null pointer");
  ncclUniqueId id;
  const ncclResult_t r = rccl().GetUniqueId(&id);
  if (r != ncclSuccess) return fail("vj_comm_unique_id", r);
  std::memcpy(id_out, (const void*)&id, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

extern "C" int vj_comm_init(vj_comm_t* comm_out, int rank, int world, const void* id) {
  VJ_NEED_RCCL("vj_comm_init");
  VJ_CHECK_ARG(comm_out != nullptr && id != nullptr, "vj_comm_init: null pointer");
  VJ_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "vj_comm_init: rank %d outside world %d", rank, world);
  ncclUniqueId uid;
  std::memcpy((void*)&uid, id, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t c = nullptr;
  const ncclResult_t r = rccl().CommInitRank(&c, world, uid, rank);   // binds to the calling thread's current HIP device
  if (r != ncclSuccess) return fail("vj_comm_init", r);
  *comm_out = (vj_comm_t)c;
  return 0;
}

extern "C" int vj_comm_allreduce_bucket(vj_comm_t comm, float* grad, int64_t count, hipStream_t stream) {
  VJ_NEED_RCCL("vj_comm_allreduce_bucket");
  VJ_CHECK_ARG(comm != nullptr && (grad != nullptr || count == 0) && count >= 0, "vj_comm_allreduce_bucket: bad arguments");
  if (count == 0) return 0;
  const ncclResult_t r = rccl().AllReduce(grad, grad, (size_t)count, ncclFloat32, ncclSum, (ncclComm_t)comm, stream);
  if (r != ncclSuccess) return fail("vj_comm_allreduce_bucket", r);
  return 0;
}

extern "C" int vj_comm_broadcast(vj_comm_t comm, float* buf, int64_t count, int root, hipStream_t stream) {
  VJ_NEED_RCCL("vj_comm_broadcast");
  VJ_CHECK_ARG(comm != nullptr && (buf != nullptr || count == 0) && count >= 0 && root >= 0, "vj_comm_broadcast: bad arguments");
  if (count == 0) return 0;
  const ncclResult_t r = rccl().Broadcast(buf, buf, (size_t)count, ncclFloat32, root, (ncclComm_t)comm, stream);
  if (r != ncclSuccess) return fail("vj_comm_broadcast", r);
  return 0;
}

extern "C" int vj_comm_destroy(vj_comm_t comm) {
  VJ_NEED_RCCL("vj_comm_destroy");
  if (comm == nullptr) return 0;
  const ncclResult_t r = rccl().CommDestroy((ncclComm_t)comm);
  if (r != ncclSuccess) return fail("vj_comm_destroy", r);
  return 0;
}
