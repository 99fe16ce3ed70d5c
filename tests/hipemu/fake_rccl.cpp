// fake_rccl.cpp — TEST INFRASTRUCTURE: the slice of RCCL that csrc/eh_comm.h binds (ncclGetUniqueId, ncclCommInitRank,
// ncclCommInitAll, ncclBroadcast, ncclAllGather, group calls), for CPU ranks.  With the emulator build of the engine "device memory"
// is host memory, so a collective is a copy through a POSIX shared-memory segment named after the unique id (ranks = OS
// processes, as under torch.distributed.run) or a plain memcpy between the buffers of one process (ncclCommInitAll).
// Loaded through EH_RCCL_LIB=build/libfake_rccl.so by tests/test_comm_abi.py; never shipped, never loaded on a GPU box.
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <vector>

namespace {
constexpr size_t CHUNK = 4u << 20;
struct Shared {
  std::atomic<uint32_t> arrived, sense;
  std::atomic<uint32_t> joined;
  uint8_t buf[CHUNK * 16];          // up to 16 ranks of one CHUNK each (all-gather), or one CHUNK (broadcast)
};
struct Comm {
  int rank, n;
  Shared* sh;                        // multi-process
  uint32_t my_sense;
  char name[136];
  struct Local* local;               // single-process group (ncclCommInitAll)
};
struct Op { int kind; const void* send; void* recv; size_t bytes; int root; Comm* c; };
struct Local { std::vector<Op> pending; int n; };
thread_local int g_group = 0;
thread_local std::vector<Op> g_ops;

size_t dsize(int dt) { return dt == 5 || dt == 4 || dt == 8 ? 8 : (dt == 2 || dt == 3 || dt == 7 ? 4 : (dt == 6 ? 2 : 1)); }

void barrier(Comm* c) {
  Shared* s = c->sh;
  c->my_sense ^= 1u;
  if (s->arrived.fetch_add(1) + 1 == (uint32_t)c->n) { s->arrived.store(0); s->sense.store(c->my_sense); }
  else { int spins = 0; while (s->sense.load() != c->my_sense) { if (++spins > 1000) usleep(50); } }
}

int run_local(std::vector<Op>& ops) {          // every rank of a one-process communicator has made its call: resolve by copies
  for (size_t i = 0; i < ops.size(); i++) {
    Op& o = ops[i];
    if (o.kind < 0) continue;
    std::vector<Op*> same;
    for (size_t j = i; j < ops.size(); j++) if (ops[j].kind == o.kind && ops[j].c->local == o.c->local && ops[j].bytes == o.bytes && ops[j].root == o.root && (j == i || ops[j].kind >= 0)) {
      bool dup = false; for (Op* q : same) if (q->c == ops[j].c) dup = true;
      if (!dup) same.push_back(&ops[j]);
    }
    if ((int)same.size() != o.c->n) return 5;  // ncclInvalidUsage: not every rank called
    if (o.kind == 0) {
      const void* src = nullptr; for (Op* q : same) if (q->c->rank == o.root) src = q->send;
      for (Op* q : same) if (q->recv != src) memmove(q->recv, src, o.bytes);
    } else {
      for (Op* q : same) for (Op* r : same) { uint8_t* dst = (uint8_t*)q->recv + (size_t)r->c->rank * o.bytes; if (dst != r->send) memmove(dst, r->send, o.bytes); }
    }
    for (Op* q : same) q->kind = -1;
  }
  return 0;
}

int do_op(Op o) {
  Comm* c = o.c;
  if (c->local) { g_ops.push_back(o); if (!g_group) { int r = c->n == 1 ? run_local(g_ops) : 5; g_ops.clear(); return r; } return 0; }
  if (c->n == 1) { if (o.kind == 0) { if (o.recv != o.send) memmove(o.recv, o.send, o.bytes); } else if (o.recv != o.send) memmove(o.recv, o.send, o.bytes); return 0; }
  for (size_t done = 0; done < o.bytes || (o.bytes == 0 && done == 0); done += CHUNK) {
    size_t k = o.bytes - done < CHUNK ? o.bytes - done : CHUNK;
    if (o.kind == 0) {
      if (c->rank == o.root) memcpy(c->sh->buf, (const uint8_t*)o.send + done, k);
      barrier(c);
      if (c->rank != o.root) memcpy((uint8_t*)o.recv + done, c->sh->buf, k); else if (o.recv != o.send) memmove((uint8_t*)o.recv + done, (const uint8_t*)o.send + done, k);
      barrier(c);
    } else {
      memcpy(c->sh->buf + (size_t)c->rank * CHUNK, (const uint8_t*)o.send + done, k);
      barrier(c);
      for (int r = 0; r < c->n; r++) { uint8_t* dst = (uint8_t*)o.recv + (size_t)r * o.bytes + done; const uint8_t* src = c->sh->buf + (size_t)r * CHUNK; if (r != c->rank || dst != (const uint8_t*)o.send + done) memcpy(dst, src, k); }
      barrier(c);
    }
    if (o.bytes == 0) break;
  }
  return 0;
}
}  // namespace

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  unsigned v[4]; FILE* f = fopen("/dev/urandom", "rb"); if (!f || fread(v, sizeof(v), 1, f) != 1) { v[0] = (unsigned)getpid(); v[1] = v[2] = v[3] = 12345; } if (f) fclose(f);
  snprintf(id->internal, sizeof(id->internal), "/eh_fake_rccl_%08x%08x%08x%08x", v[0], v[1], v[2], v[3]);
  return 0;
}
int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > 16 || rank < 0 || rank >= nranks) return 4;
  Comm* c = new Comm(); c->rank = rank; c->n = nranks; c->my_sense = 0; c->local = nullptr; c->sh = nullptr;
  snprintf(c->name, sizeof(c->name), "%s", id.internal);
  if (nranks > 1) {
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) { delete c; return 2; }
    c->sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->sh == MAP_FAILED) { delete c; return 2; }
    c->sh->joined.fetch_add(1);
    int spins = 0; while (c->sh->joined.load() < (uint32_t)nranks) { usleep(200); if (++spins > 300000) { delete c; return 2; } }   // rendezvous (60 s)
  }
  *comm = c;
  return 0;
}
int ncclCommInitAll(void** comms, int ndev, const int*) {
  Local* lo = new Local(); lo->n = ndev;
  for (int i = 0; i < ndev; i++) { Comm* c = new Comm(); c->rank = i; c->n = ndev; c->sh = nullptr; c->my_sense = 0; c->name[0] = 0; c->local = lo; comms[i] = c; }
  return 0;
}
int ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (c->sh) { munmap(c->sh, sizeof(Shared)); shm_unlink(c->name); }
  delete c;
  return 0;
}
int ncclBroadcast(const void* send, void* recv, size_t count, int dt, int root, void* comm, void*) { Op o{0, send, recv, count * dsize(dt), root, (Comm*)comm}; return do_op(o); }
int ncclAllGather(const void* send, void* recv, size_t count, int dt, void* comm, void*) { Op o{1, send, recv, count * dsize(dt), 0, (Comm*)comm}; return do_op(o); }
int ncclGroupStart() { g_group++; return 0; }
int ncclGroupEnd() { if (--g_group == 0) { int r = run_local(g_ops); g_ops.clear(); return r; } return 0; }
const char* ncclGetErrorString(int r) { return r == 0 ? "no error" : r == 5 ? "invalid usage (fake_rccl)" : "error (fake_rccl)"; }
}
