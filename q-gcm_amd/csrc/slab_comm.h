// RCCL transport of the y-slab exchanges, driven from C++ (no Python between the kernels and the collectives).
// The reference has no decomposition: nothing here replaces reference code. RCCL is bound at run time
// (dlopen), so a single-GPU process never loads it; inside a torch process the already loaded librccl is reused.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

struct QgRccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

// returns nullptr and fills err on failure
static QgRccl *qg_rccl(char *err, size_t nerr) {
  static QgRccl R;
  if (R.lib) return &R;
  const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *n : names) // a copy that is already in the process (torch's) wins
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
  if (!h)
    for (const char *n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!h) {
    snprintf(err, nerr, "cannot load librccl: %s", dlerror());
    return nullptr;
  }
#define QG_SYM(field, name)                                                  \
  *(void **)(&R.field) = dlsym(h, name);                                     \
  if (!R.field) {                                                            \
    snprintf(err, nerr, "librccl lacks %s", name);                           \
    return nullptr;                                                          \
  }
  QG_SYM(GetUniqueId, "ncclGetUniqueId")
  QG_SYM(CommInitRank, "ncclCommInitRank")
  QG_SYM(CommDestroy, "ncclCommDestroy")
  QG_SYM(AllGather, "ncclAllGather")
  QG_SYM(Send, "ncclSend")
  QG_SYM(Recv, "ncclRecv")
  QG_SYM(GroupStart, "ncclGroupStart")
  QG_SYM(GroupEnd, "ncclGroupEnd")
  QG_SYM(GetErrorString, "ncclGetErrorString")
#undef QG_SYM
  R.lib = h;
  return &R;
}

// communicator + exchange buffers of one slab handle
struct QgSlabComm {
  QgRccl *api = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  bool halo_p2p = true; // halo rows by grouped send/recv with the two neighbours (false: one all-gather of all edge rows)
  size_t th_len = 0, halo_len = 0;
  double *th_send = nullptr, *th_gath = nullptr; // slab summaries of the y sweeps (k_thomas.h, TH_MSG per mode and wavenumber)
  double *h_send = nullptr, *h_gath = nullptr;   // edge rows: [to lower | to upper] per rank
  double *oml_send = nullptr, *oml_gath = nullptr; // mixed layer: the three sums of a slab's k_oml_step (3 per rank)
  // halo exchange overlapped with the inner tile rows of the next step's tendency launch (qgcm_hip_comm_set_overlap):
  // the exchange and the halo unpack run on cstream, forked from / joined to the handle's stream by events
  bool overlap = false, pending = false;
  bool outer_done = false; // the pending exchange's stream has also run the outer tile rows of the next step's tendency
  hipStream_t cstream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_halo = nullptr;
};
