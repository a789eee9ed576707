// slab_link.hpp -- multi-GPU behind the C ABI: an RCCL communicator handle and the slab step
// drivers (the halo / migrant exchange of a slab with its two neighbours, the split WCSPH step that
// hides it under the interior force launch, the PCISPH step with its per-iteration error
// all-reduce, the periodic re-plan of the message sizes).  Included by dslsph.hip; gfx950 / ROCm only.
//
// RCCL is bound at run time (dlopen "librccl.so.1"), not linked: a single-GPU host never loads the
// 570 MB library, and a process that already carries an RCCL of that SONAME (PyTorch bundles one)
// gets that same copy instead of a second one with its own global state.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

struct dsl_comm {
  ncclComm_t comm = nullptr;
  int nranks = 0, rank = 0, device = 0;
  bool custom = false;  // dsl_comm_create_custom: the host's transport table instead of RCCL
  dsl_transport tr{};
  std::string err;
};

namespace {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

RcclApi& rccl() {
  static RcclApi api;
  return api;
}

bool rccl_load() {
  RcclApi& a = rccl();
  if (a.lib) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (a.lib) break;
  }
  if (!a.lib) {
    a.err = std::string("cannot load RCCL: ") + (dlerror() ? dlerror() : "librccl.so.1 not found");
    return false;
  }
  bool ok = true;
  auto sym = [&](const char* s) {
    void* p = dlsym(a.lib, s);
    if (!p) {
      ok = false;
      a.err = std::string("RCCL symbol missing: ") + s;
    }
    return p;
  };
  a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
  a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
  a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(sym("ncclCommInitAll"));
  a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
  a.CommCount = reinterpret_cast<decltype(a.CommCount)>(sym("ncclCommCount"));
  a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
  a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
  a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
  a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
  a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
  a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
  if (!ok) {
    dlclose(a.lib);
    a.lib = nullptr;
  }
  return ok;
}

std::string g_comm_error;

#define NCCL_TRY_H(h, expr)                                                                             \
  do {                                                                                                  \
    ncclResult_t r__ = (expr);                                                                          \
    if (r__ != ncclSuccess) return fail((h), DSL_ERR_DEVICE, std::string(#expr) + ": " + rccl().GetErrorString(r__)); \
  } while (0)

// The slab drivers' transport calls: RCCL, or the host's table (dsl_comm_create_custom) -- the SAME call
// sequence either way, which is what lets the N > 1 protocol (message order between distinct ranks, the
// re-plan and PCISPH all-reduces) run in tests on a box where RCCL cannot (several ranks on one device).
// The all-reduce words are non-negative (status counts, the bits of a non-negative float): signed and
// unsigned MAX agree on them.
#define XFER_TRY_H(h, expr, what)                                                                        \
  do {                                                                                                   \
    if ((expr) != 0) return fail((h), DSL_ERR_DEVICE, std::string("transport callback failed: ") + (what)); \
  } while (0)
inline int xfer_group_start(dsl_handle* h, dsl_comm* c) {
  if (c->custom) {
    XFER_TRY_H(h, c->tr.group_start(c->tr.ctx), "group_start");
    return DSL_OK;
  }
  NCCL_TRY_H(h, rccl().GroupStart());
  return DSL_OK;
}
inline int xfer_group_end(dsl_handle* h, dsl_comm* c) {
  if (c->custom) {
    XFER_TRY_H(h, c->tr.group_end(c->tr.ctx), "group_end");
    return DSL_OK;
  }
  NCCL_TRY_H(h, rccl().GroupEnd());
  return DSL_OK;
}
inline int xfer_send(dsl_handle* h, dsl_comm* c, const float* buf, size_t n, int peer, hipStream_t st) {
  if (c->custom) {
    XFER_TRY_H(h, c->tr.send(c->tr.ctx, buf, n * sizeof(float), peer, (void*)st), "send");
    return DSL_OK;
  }
  NCCL_TRY_H(h, rccl().Send(buf, n, ncclFloat, peer, c->comm, st));
  return DSL_OK;
}
inline int xfer_recv(dsl_handle* h, dsl_comm* c, float* buf, size_t n, int peer, hipStream_t st) {
  if (c->custom) {
    XFER_TRY_H(h, c->tr.recv(c->tr.ctx, buf, n * sizeof(float), peer, (void*)st), "recv");
    return DSL_OK;
  }
  NCCL_TRY_H(h, rccl().Recv(buf, n, ncclFloat, peer, c->comm, st));
  return DSL_OK;
}
inline int xfer_all_reduce_max(dsl_handle* h, dsl_comm* c, void* words, size_t count, hipStream_t st) {
  if (c->custom) {
    XFER_TRY_H(h, c->tr.all_reduce_max_u32(c->tr.ctx, words, count, (void*)st), "all_reduce_max_u32");
    return DSL_OK;
  }
  NCCL_TRY_H(h, rccl().AllReduce(words, words, count, ncclUint32, ncclMax, c->comm, st));
  return DSL_OK;
}

}  // namespace

// The slab's link to its neighbours (dsl_slab_attach): communicator, neighbour ranks, message
// buffers, the side stream the transfer runs on.
struct SlabLink {
  dsl_comm* comm = nullptr;
  int lo = -1, hi = -1;  // neighbour ranks; -1 = domain end
  float width_full = 0.f, width = 0.f;
  int cap_full = 0, cap_x = 0, max_full = 0, max_x = 0;
  bool overlap = false, ghosts_in = false;
  float* send[2] = {nullptr, nullptr};
  float* recv[2] = {nullptr, nullptr};
  size_t buf_floats = 0;
  int* dev_words = nullptr;  // 4 ints for the re-plan all-reduce
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_pack = nullptr, ev_xfer = nullptr;
  int64_t steps = 0;
  int replan_every = 8;
  // periodic images along the slab axis: records arriving from the lower / upper neighbour are moved by
  // these amounts (a ring closes with -L / +L at its two end ranks; 0 elsewhere)
  float shift_from_lo = 0.f, shift_from_hi = 0.f;
};
