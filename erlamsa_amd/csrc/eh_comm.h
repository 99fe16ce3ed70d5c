// eh_comm.h — the seed arena on every GPU of the node, over RCCL called from inside the library (host side).
//
// Cases are independent (erlamsa_main.erl:179-183: the worker is re-seeded per case), so the mutation path never communicates.
// The one exchange is the arena at load time: a broadcast from the rank that holds it (BASELINE configs[3]) or an all-gather of
// per-rank shards (configs[4]: 8 x 8 GiB, every link carries 1/8 instead of one root feeding seven peers).  The reference's own
// counterpart is `--workers` reading the same files on every scheduler (erlamsa_main.erl:90-108); RCCL is reached from here, not
// from the host, because the host north_star names - the BEAM - has no HIP or RCCL binding: it hands 128 bytes of unique id
// from rank 0 to the other OS processes over Erlang distribution and calls the entry points below.
// librccl.so is loaded on first use (EH_RCCL_LIB names another file - tests/hipemu/fake_rccl.cpp on the CPU emulator), so a
// single-GPU host needs no RCCL at all.
#pragma once
#include <dlfcn.h>

namespace ehcomm {

typedef struct { char internal[128]; } UniqueId;       // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128, rccl.h:40-43)
typedef void* Comm;                                     // ncclComm_t
enum { kUint8 = 1, kUint64 = 5 };                       // ncclDataType_t (rccl.h:459-464)

struct Api {
  void* h = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommInitAll)(Comm*, int, const int*) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string err;
};

inline Api* api() {
  static Api a;
  static std::mutex m;
  std::lock_guard<std::mutex> g(m);
  if (a.h) return &a;
  const char* name = getenv("EH_RCCL_LIB");
  void* h = nullptr;
  if (!name || !*name) {
    // a host that has RCCL in the process already (torch brings its own copy) shares that instance; otherwise the system's
    name = "librccl.so";
    h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  } else h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
  if (!h) { const char* de = dlerror(); a.err = std::string("cannot load RCCL (") + name + "): " + (de ? de : "?"); return &a; }   // (dlerror() hands its text out once)
  bool ok = true;
  auto sym = [&](const char* s) -> void* { void* p = dlsym(h, s); if (!p) { ok = false; a.err = std::string("RCCL symbol missing: ") + s; } return p; };
  a.GetUniqueId = (int (*)(UniqueId*))sym("ncclGetUniqueId");
  a.CommInitRank = (int (*)(Comm*, int, UniqueId, int))sym("ncclCommInitRank");
  a.CommInitAll = (int (*)(Comm*, int, const int*))sym("ncclCommInitAll");
  a.CommDestroy = (int (*)(Comm))sym("ncclCommDestroy");
  a.Broadcast = (int (*)(const void*, void*, size_t, int, int, Comm, hipStream_t))sym("ncclBroadcast");
  a.AllGather = (int (*)(const void*, void*, size_t, int, Comm, hipStream_t))sym("ncclAllGather");
  a.GroupStart = (int (*)())sym("ncclGroupStart");
  a.GroupEnd = (int (*)())sym("ncclGroupEnd");
  a.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
  if (ok) a.h = h;
  return &a;
}

}  // namespace ehcomm
