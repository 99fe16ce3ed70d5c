// TEST INFRASTRUCTURE — a CPU stand-in for <hip/hip_runtime.h> that runs the UNMODIFIED engine sources
// (erlamsa_amd/csrc/*.h, eh_engine.hip) on the host, one wavefront at a time.  Never shipped, never
// loaded by the product package; tests/test_emulated_kernel.py builds build/liberlamsa_hip_emu.so with
//   g++ -x c++ -I tests/hipemu -I include erlamsa_amd/csrc/eh_engine.hip
// and runs small parity cases through the same C ABI.  Purpose: kernel logic (wave-uniform control
// flow, cross-lane exchanges, work-area bookkeeping) can be checked without GPU minutes, and a cross-lane
// operation reached by only part of the wavefront is reported instead of silently mis-executing.
//
// Model: a workgroup of 64 lanes = 64 ucontext fibers on one OS thread.  A fiber runs until it reaches a
// cross-lane operation (ballot, shuffle, readlane, ds_permute, __syncthreads), deposits its operand and
// yields; when every live lane has arrived at the same operation the exchange is resolved and all lanes
// continue.  Memory is ordinary host memory; lanes run one after another between rendezvous points, so
// code that relies on lockstep execution without a wave_sync()/cross-lane operation between a store and
// another lane's load of it fails here (it would be a race on the GPU as well).  The converse - a load that only works because
// a LOWER lane stored first - is what build_emu.py --race + race_hooks.cpp catch (every load / store instrumented).
#pragma once
#include <math.h>
#include <ucontext.h>
#include <execinfo.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

// ---- qualifiers ------------------------------------------------------------------------------
#define __device__
#define __global__
#define __host__
// __shared__ objects are collected in one ELF section.  The engine only ever stores wave-uniform values
// to LDS (every lane writes the same value), which on the GPU happens in lockstep; fibers run one after
// another, so `x += n` executed by 64 lanes in turn would add 64 n.  Each lane therefore gets its own
// copy of the section (swapped in and out on every fiber switch), and at every rendezvous the copies are
// compared: a difference means some lane stored a non-uniform value to LDS, i.e. a real bug.
#define __shared__ __attribute__((section("hipemu_shared"), used))
extern "C" char __start_hipemu_shared[], __stop_hipemu_shared[];
#define __constant__
#define HIPEMU 1        // lets the engine place lane-written LDS scratch (EH_LDS_ARRAY) in plain memory: the LDS model here is per lane
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define HIP_SYMBOL(x) (&(x))

struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct hipemu_idx { unsigned x, y, z; };
inline hipemu_idx threadIdx{0, 0, 0}, blockIdx{0, 0, 0}, blockDim{64, 1, 1}, gridDim{1, 1, 1};

// ---- race detector hooks (build_emu.py --race links tests/hipemu/race_hooks.cpp; otherwise nothing) -------------------------
#ifdef HIPEMU_RACE
extern "C" { void hipemu_race_init(void); void hipemu_race_exclude(const void* lo, const void* hi); void hipemu_race_clear_excludes(void);
             void hipemu_race_lane(int lane); void hipemu_race_epoch(void); unsigned long hipemu_race_count(void); void hipemu_race_atomic(int on); void hipemu_race_region(const void* p, size_t n, int add); }
#define HIPEMU_RACE_CALL(x) x
#else
#define HIPEMU_RACE_CALL(x) ((void)0)
#endif

// ---- the wavefront emulator --------------------------------------------------------------------
namespace hipemu {
constexpr int W = 64;
constexpr size_t STACK = 1u << 20;
enum Op { OP_NONE, OP_BALLOT, OP_SHFL, OP_SHFL_UP, OP_SHFL_DOWN, OP_SHFL_XOR, OP_READLANE, OP_READFIRST, OP_PERMUTE, OP_BPERMUTE, OP_SYNC };
enum State { RUN, WAIT, DONE };
struct Wave {
  ucontext_t sched;
  ucontext_t ctx[W];
  char* stacks = nullptr;
  int state[W];
  int op[W];
  unsigned opcount[W];
  int cur = 0;
  uint64_t val[2][W];
  int64_t arg[2][W];
  uint8_t present[2][W];
  const std::function<void()>* body = nullptr;
  char* lds_save = nullptr;       // W copies of the hipemu_shared section
  size_t lds_size = 0;
};
inline Wave g_wave;

inline void trampoline() {
  Wave& w = g_wave;
  (*w.body)();
  int l = w.cur;
  w.state[l] = DONE;
  swapcontext(&w.ctx[l], &w.sched);
}

// deposit and wait for the rest of the wavefront; returns the buffer parity the operands are in
inline unsigned rendezvous(int op, uint64_t v, int64_t a) {
  Wave& w = g_wave;
  int l = w.cur;
  unsigned par = w.opcount[l] & 1u;
  w.val[par][l] = v; w.arg[par][l] = a; w.present[par][l] = 1; w.op[l] = op; w.state[l] = WAIT;
  swapcontext(&w.ctx[l], &w.sched);
  w.opcount[l]++;
  return par;
}

inline void run_block(const std::function<void()>& body) {
  Wave& w = g_wave;
  if (!w.stacks) w.stacks = (char*)malloc(STACK * W);
  w.body = &body;
  memset(w.present, 0, sizeof(w.present));
  w.lds_size = (size_t)(__stop_hipemu_shared - __start_hipemu_shared);
  if (!w.lds_save) w.lds_save = (char*)malloc(w.lds_size * W + 1);
  memset(w.lds_save, 0, w.lds_size * W);
  HIPEMU_RACE_CALL(hipemu_race_init());
  HIPEMU_RACE_CALL(hipemu_race_clear_excludes());
  HIPEMU_RACE_CALL(hipemu_race_exclude(w.stacks, w.stacks + STACK * W));                   // private: the fibers' stacks
  HIPEMU_RACE_CALL(hipemu_race_exclude(__start_hipemu_shared, __stop_hipemu_shared));      // LDS: one image per lane, compared at every rendezvous
  HIPEMU_RACE_CALL(hipemu_race_exclude(&w, &w + 1));                                       // the emulator's own exchange buffers
  HIPEMU_RACE_CALL(hipemu_race_epoch());
  for (int l = 0; l < W; l++) {
    getcontext(&w.ctx[l]);
    w.ctx[l].uc_stack.ss_sp = w.stacks + STACK * l;
    w.ctx[l].uc_stack.ss_size = STACK;
    w.ctx[l].uc_link = nullptr;
    makecontext(&w.ctx[l], (void (*)())trampoline, 0);
    w.state[l] = RUN; w.op[l] = OP_NONE; w.opcount[l] = 0;
  }
  while (true) {
    for (int l = 0; l < W; l++) {
      if (w.state[l] != RUN) continue;
      w.cur = l; threadIdx.x = (unsigned)l;
      memcpy(__start_hipemu_shared, w.lds_save + w.lds_size * l, w.lds_size);
      HIPEMU_RACE_CALL(hipemu_race_lane(l));
      swapcontext(&w.sched, &w.ctx[l]);
      HIPEMU_RACE_CALL(hipemu_race_lane(-1));
      memcpy(w.lds_save + w.lds_size * l, __start_hipemu_shared, w.lds_size);
    }
    int waiting = 0, op = OP_NONE; unsigned cnt = 0; bool mixed = false;
    for (int l = 0; l < W; l++) {
      if (w.state[l] != WAIT) continue;
      if (!waiting) { op = w.op[l]; cnt = w.opcount[l]; }
      else if (w.op[l] != op || w.opcount[l] != cnt) mixed = true;
      waiting++;
    }
    if (!waiting) break;                                    // every lane returned from the kernel
    if (mixed) {
      fprintf(stderr, "hipemu: lanes of block %u reached different cross-lane operations (divergent wave op):", blockIdx.x);
      for (int l = 0; l < W; l++) if (w.state[l] == WAIT) fprintf(stderr, " %d:%d/%u", l, w.op[l], w.opcount[l]);
      fprintf(stderr, "\n");
      abort();
    }
    {                                                       // LDS must hold wave-uniform values only
      int first = -1;
      for (int l = 0; l < W; l++) {
        if (w.state[l] != WAIT) continue;
        if (first < 0) { first = l; continue; }
        const char* a = w.lds_save + w.lds_size * first; const char* b = w.lds_save + w.lds_size * l;
        for (size_t o = 0; o < w.lds_size;) {
          if (a[o] == b[o]) { o++; continue; }
          // a pointer into the lane's own stack (address of a by-value kernel argument) differs between
          // fibers by construction; it is the same private address on every lane of a real wavefront
          size_t o8 = o & ~(size_t)7; uintptr_t pa, pb; memcpy(&pa, a + o8, 8); memcpy(&pb, b + o8, 8);
          uintptr_t sa = (uintptr_t)(w.stacks + STACK * first), sb = (uintptr_t)(w.stacks + STACK * l);
          if (pa >= sa && pa < sa + STACK && pb >= sb && pb < sb + STACK && pa - sa == pb - sb) { o = o8 + 8; continue; }
          fprintf(stderr, "hipemu: non-uniform LDS contents at a rendezvous (block %u, lanes %d and %d, section offset %zu, op %d)\n", blockIdx.x, first, l, o, op);
          abort();
        }
      }
    }
    unsigned par = cnt & 1u;
    for (int l = 0; l < W; l++) w.present[par ^ 1u][l] = 0;   // the other buffer has been read by everybody
    for (int l = 0; l < W; l++) if (w.state[l] == WAIT) w.state[l] = RUN;
    HIPEMU_RACE_CALL(hipemu_race_epoch());                   // a rendezvous orders what came before it against what comes after
  }
}

template <class F>
inline void launch(dim3 grid, dim3 block, F&& f) {
  if (block.x != (unsigned)W || block.y != 1 || block.z != 1) { fprintf(stderr, "hipemu: workgroup must be one 64-lane wavefront\n"); abort(); }
  std::function<void()> body = f;
  gridDim = {grid.x, 1, 1};
  for (unsigned b = 0; b < grid.x; b++) { blockIdx.x = b; run_block(body); }
}
}  // namespace hipemu

// ---- cross-lane operations -----------------------------------------------------------------------
inline unsigned long long __ballot(int pred) {
  unsigned p = hipemu::rendezvous(hipemu::OP_BALLOT, pred ? 1 : 0, 0);
  unsigned long long m = 0;
  for (int l = 0; l < hipemu::W; l++) if (hipemu::g_wave.present[p][l] && hipemu::g_wave.val[p][l]) m |= 1ull << l;
  return m;
}
inline int hipemu_fetch(unsigned p, int src, int own) { return (src >= 0 && src < hipemu::W && hipemu::g_wave.present[p][src]) ? (int)(uint32_t)hipemu::g_wave.val[p][src] : own; }
inline int __shfl(int v, int src) { unsigned p = hipemu::rendezvous(hipemu::OP_SHFL, (uint32_t)v, src); return hipemu_fetch(p, src & 63, v); }
inline int __shfl_up(int v, unsigned d) { unsigned p = hipemu::rendezvous(hipemu::OP_SHFL_UP, (uint32_t)v, d); int s = (int)threadIdx.x - (int)d; return s < 0 ? v : hipemu_fetch(p, s, v); }
inline int __shfl_down(int v, unsigned d) { unsigned p = hipemu::rendezvous(hipemu::OP_SHFL_DOWN, (uint32_t)v, d); int s = (int)threadIdx.x + (int)d; return s >= hipemu::W ? v : hipemu_fetch(p, s, v); }
inline int __shfl_xor(int v, int m) { unsigned p = hipemu::rendezvous(hipemu::OP_SHFL_XOR, (uint32_t)v, m); int s = (int)threadIdx.x ^ m; return (s < 0 || s >= hipemu::W) ? v : hipemu_fetch(p, s, v); }
// v_readlane_b32 takes its lane select from a scalar register: the index must be the same on every lane (use __shfl
// for a per-lane source).  A non-uniform index "works" lane by lane on a CPU, so it is checked here.
inline int __builtin_amdgcn_readlane(int v, int lane) {
  unsigned p = hipemu::rendezvous(hipemu::OP_READLANE, (uint32_t)v, lane);
  for (int l = 0; l < hipemu::W; l++)
    if (hipemu::g_wave.present[p][l] && (int)hipemu::g_wave.arg[p][l] != lane) { fprintf(stderr, "hipemu: v_readlane with a non-uniform lane index (block %u, lanes %d and %d: %d vs %d)\n", blockIdx.x, (int)threadIdx.x, l, lane, (int)hipemu::g_wave.arg[p][l]); void* bt[16]; int nb = backtrace(bt, 16); backtrace_symbols_fd(bt, nb, 2); abort(); }
  return hipemu_fetch(p, lane & 63, v);
}
inline int __builtin_amdgcn_readfirstlane(int v) {
  unsigned p = hipemu::rendezvous(hipemu::OP_READFIRST, (uint32_t)v, 0);
  for (int l = 0; l < hipemu::W; l++) if (hipemu::g_wave.present[p][l]) return (int)(uint32_t)hipemu::g_wave.val[p][l];
  return v;
}
// forward permute: lane i sends v to lane (addr >> 2) & 63; a lane nobody wrote to reads 0
inline int __builtin_amdgcn_ds_permute(int addr, int v) {
  unsigned p = hipemu::rendezvous(hipemu::OP_PERMUTE, (uint32_t)v, addr);
  int me = (int)threadIdx.x, r = 0;
  for (int l = 0; l < hipemu::W; l++) if (hipemu::g_wave.present[p][l] && (int)((hipemu::g_wave.arg[p][l] >> 2) & 63) == me) r = (int)(uint32_t)hipemu::g_wave.val[p][l];
  return r;
}
inline int __builtin_amdgcn_ds_bpermute(int addr, int v) { unsigned p = hipemu::rendezvous(hipemu::OP_BPERMUTE, (uint32_t)v, addr); int s = (addr >> 2) & 63; return hipemu::g_wave.present[p][s] ? (int)(uint32_t)hipemu::g_wave.val[p][s] : 0; }
inline void __syncthreads() { (void)hipemu::rendezvous(hipemu::OP_SYNC, 0, 0); }

using std::isinf;
using std::isnan;
// ---- scalar helpers ------------------------------------------------------------------------------
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline long long __double_as_longlong(double d) { long long r; memcpy(&r, &d, 8); return r; }
inline double __longlong_as_double(long long v) { double r; memcpy(&r, &v, 8); return r; }
inline unsigned long long __builtin_readcyclecounter() {
  return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
template <class T, class U> inline T atomicAdd(T* p, U v) { HIPEMU_RACE_CALL(hipemu_race_atomic(1)); T old = *p; *p = (T)(old + (T)v); HIPEMU_RACE_CALL(hipemu_race_atomic(0)); return old; }
template <class T, class U> inline T atomicExch(T* p, U v) { HIPEMU_RACE_CALL(hipemu_race_atomic(1)); T old = *p; *p = (T)v; HIPEMU_RACE_CALL(hipemu_race_atomic(0)); return old; }
template <class T, class U> inline T atomicOr(T* p, U v) { HIPEMU_RACE_CALL(hipemu_race_atomic(1)); T old = *p; *p = (T)(old | (T)v); HIPEMU_RACE_CALL(hipemu_race_atomic(0)); return old; }
template <class T, class U> inline T atomicMin(T* p, U v) { HIPEMU_RACE_CALL(hipemu_race_atomic(1)); T old = *p; if ((T)v < old) *p = (T)v; HIPEMU_RACE_CALL(hipemu_race_atomic(0)); return old; }
template <class T, class U> inline T atomicMax(T* p, U v) { HIPEMU_RACE_CALL(hipemu_race_atomic(1)); T old = *p; if ((T)v > old) *p = (T)v; HIPEMU_RACE_CALL(hipemu_race_atomic(0)); return old; }
inline unsigned long long __builtin_amdgcn_s_memrealtime() { return __builtin_readcyclecounter() / 10; }
template <class T, class U, class V> inline T atomicCAS(T* p, U cmp, V v) { HIPEMU_RACE_CALL(hipemu_race_atomic(1)); T old = *p; if (old == (T)cmp) *p = (T)v; HIPEMU_RACE_CALL(hipemu_race_atomic(0)); return old; }

// ---- the slice of the HIP runtime API the engine's host side uses -----------------------------------
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorOutOfMemory = 2;
constexpr hipError_t hipErrorNotReady = 600;
typedef void* hipStream_t;
struct hipemu_event { std::chrono::steady_clock::time_point t; };
typedef hipemu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; size_t totalGlobalMem; char gcnArchName[256]; };

inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "hipemu: out of memory"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { const char* e = getenv("HIPEMU_DEVICES"); *n = e && atoi(e) > 0 ? atoi(e) : 1; return hipSuccess; }   // (HIPEMU_DEVICES: several "devices", all of them this CPU - the multi-GPU entry points of the C ABI, tests/test_comm_abi.py)
inline hipError_t hipSetDevice(int) { return hipSuccess; }
enum hipLimit_t { hipLimitStackSize = 0 };
inline hipError_t hipDeviceSetLimit(hipLimit_t, size_t) { return hipSuccess; }   // fibers have 1 MiB stacks
inline hipError_t hipDeviceGetLimit(size_t* v, hipLimit_t) { *v = (size_t)1 << 20; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof(*p)); strcpy(p->name, "hipemu (CPU wavefront emulator)"); strcpy(p->gcnArchName, "hipemu");
  p->multiProcessorCount = 1; p->totalGlobalMem = (size_t)8 << 30;
  return hipSuccess;
}
#ifdef HIPEMU_RACE
constexpr size_t HIPEMU_GUARD = 256;   // race build: a guard zone on either side of every device allocation (race_hooks.cpp reports accesses to it)
#else
constexpr size_t HIPEMU_GUARD = 0;
#endif
template <class T> inline hipError_t hipMalloc(T** p, size_t n) {
  void* q = nullptr;
  if (posix_memalign(&q, 256, (n ? n : 256) + 2 * HIPEMU_GUARD) != 0) { *p = nullptr; return hipErrorOutOfMemory; }
  *p = (T*)((char*)q + HIPEMU_GUARD); HIPEMU_RACE_CALL(hipemu_race_region((char*)q + HIPEMU_GUARD, n ? n : 256, 1)); return hipSuccess;
}
inline hipError_t hipFree(void* p) { if (!p) return hipSuccess; HIPEMU_RACE_CALL(hipemu_race_region(p, 0, 0)); free((char*)p - HIPEMU_GUARD); return hipSuccess; }
#define hipHostMallocDefault 0
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
inline hipError_t hipMemcpyToSymbol(void* sym, const void* s, size_t n) { memcpy(sym, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
#define hipStreamNonBlocking 1
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline void __threadfence() {}
inline hipError_t hipMemGetInfo(size_t* fr, size_t* tot) { *fr = *tot = (size_t)4 << 30; return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }   // (launches are synchronous here: whatever was recorded has happened)
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipemu::launch(dim3(grid), dim3(block), [=]() { kernel(__VA_ARGS__); })
