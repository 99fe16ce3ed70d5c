// TEST INFRASTRUCTURE - the race detector of the CPU wavefront emulator (tests/hipemu/hip/hip_runtime.h).
//
// build_emu.py --race compiles the engine with gcc's outline address instrumentation (-fsanitize=kernel-address with a call
// threshold of 0: every load and store of the kernel code becomes a call to __asan_loadN_noabort / __asan_storeN_noabort) and links
// THIS file, which implements those calls - not as an address sanitizer but as a check of the one rule the emulator cannot see by
// running lanes one after another: between two rendezvous points (cross-lane operations, wave_sync) no lane may read a byte that
// ANOTHER lane wrote, and no lane may overwrite a byte another lane read.  The emulator runs lane 0 to its next rendezvous, then
// lane 1, ...: a reader with a higher lane number sees the write, one with a lower number does not, and a real wavefront promises
// neither.  Every byte remembers (epoch, lane) of its last write and last read; a conflict inside one epoch is reported with a
// backtrace and counted (hipemu_race_count, read by the test).  The same hooks check BOUNDS: the emulator's hipMalloc / hipFree
// register the device allocations, and an access that falls just outside one of them (and inside none) is reported
// (hipemu_oob_count) - on the GPU such an access can fault the device, on the CPU it silently reads malloc's neighbourhood.  Same-lane accesses, accesses from the scheduler (no lane running),
// fiber stacks, the per-lane LDS image and the emulated atomics (and plain reads of what an atomic wrote: the engine reads its
// bitmaps with relaxed atomic loads) are not checked.  Compiled WITHOUT instrumentation.
#include <execinfo.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace {
struct Ent { uintptr_t addr; uint32_t w_epoch, r_epoch; int16_t w_lane, r_lane; };
constexpr size_t TBITS = 23, TSIZE = (size_t)1 << TBITS;
Ent* g_tab = nullptr;
uint32_t g_epoch = 1;
int g_lane = -1;
int g_atomic = 0;                         // inside an emulated atomic: its accesses are ordered by the hardware, and what it wrote may be read by anybody
unsigned long g_races = 0;
uintptr_t g_ex_lo[4], g_ex_hi[4]; int g_nex = 0;
// device allocations (the emulator's hipMalloc / hipFree, which in this build put a guard zone of OOB_NEAR bytes on either side of
// every allocation): an access inside a guard zone is an overrun - on the GPU a fault that can take the box down, on the CPU
// usually a silent read of malloc's neighbourhood
struct Region { uintptr_t lo, hi; };
Region g_reg[512]; int g_nreg = 0;
constexpr uintptr_t OOB_NEAR = 256;
unsigned long g_oob = 0;

inline Ent* slot(uintptr_t a) {
  size_t h = (size_t)((a * 0x9E3779B97F4A7C15ull) >> (64 - TBITS));
  for (size_t k = 0; k < 64; k++) {
    Ent* e = &g_tab[(h + k) & (TSIZE - 1)];
    if (e->addr == a) return e;
    if (e->addr == 0 || (e->w_epoch != g_epoch && e->r_epoch != g_epoch)) { e->addr = a; e->w_epoch = e->r_epoch = 0; e->w_lane = e->r_lane = -1; return e; }   // empty or stale: take it
  }
  return nullptr;                                                              // neighbourhood full of live entries: this byte goes unchecked
}
// one report per call site (the instruction that made the access), with a count
struct Site { void* pc; unsigned long n; };
Site g_sites[256]; int g_nsites = 0;
__attribute__((noinline)) void report_oob(uintptr_t a, size_t n, bool store, const Region& r) {
  g_oob++;
  void* bt[12]; int k = backtrace(bt, 12);
  void* pc = k > 2 ? bt[2] : nullptr;
  static void* seen[64]; static int nseen = 0;
  for (int i = 0; i < nseen; i++) if (seen[i] == pc) return;
  if (nseen < 64) seen[nseen++] = pc;
  if (getenv("HIPEMU_RACE_QUIET")) return;
  fprintf(stderr, "hipemu out of bounds: lane %d %s %zu byte(s) at %p, %ld bytes %s the device allocation [%p, %p)\n", g_lane, store ? "stores" : "loads", n, (void*)a,
          (long)(a >= r.hi ? a - r.hi : r.lo - a), a >= r.hi ? "behind" : "in front of", (void*)r.lo, (void*)r.hi);
  backtrace_symbols_fd(bt + 2, k > 2 ? (k - 2 < 5 ? k - 2 : 5) : 0, 2);
}
__attribute__((noinline)) void report(const char* what, uintptr_t a, int other) {
  g_races++;
  void* bt[12]; int n = backtrace(bt, 12);
  void* pc = n > 2 ? bt[2] : nullptr;                                           // report <- __asan_*_noabort (access is inlined) <- the kernel code
  for (int i = 0; i < g_nsites; i++) if (g_sites[i].pc == pc) { g_sites[i].n++; return; }
  if (g_nsites < 256) { g_sites[g_nsites].pc = pc; g_sites[g_nsites].n = 1; g_nsites++; }
  if (getenv("HIPEMU_RACE_QUIET")) return;
  fprintf(stderr, "hipemu race: lane %d %s byte %p that lane %d %s since the last rendezvous\n", g_lane, what, (void*)a, other, what[0] == 'r' ? "wrote" : "read");
  backtrace_symbols_fd(bt + 2, n > 2 ? (n - 2 < 5 ? n - 2 : 5) : 0, 2);
}
void report_oob(uintptr_t a, size_t n, bool store, const Region& r);
__attribute__((always_inline)) inline void access(uintptr_t a, size_t n, bool store) {
  if (g_lane < 0 || !g_tab) return;
  {                                                                              // bounds: inside a device allocation, or nowhere near one
    const Region* near = nullptr; bool inside = false;
    for (int i = 0; i < g_nreg; i++) {
      const Region& r = g_reg[i];
      if (a >= r.lo && a + n <= r.hi) { inside = true; break; }
      if (a + n > r.lo - OOB_NEAR && a < r.hi + OOB_NEAR) near = &r;
    }
    if (!inside && near) report_oob(a, n, store, *near);
  }
  for (int i = 0; i < g_nex; i++) if (a >= g_ex_lo[i] && a < g_ex_hi[i]) return;
  for (size_t i = 0; i < n; i++) {
    Ent* e = slot(a + i);
    if (!e) continue;
    if (g_atomic) { if (store) { e->w_epoch = g_epoch; e->w_lane = -2; } continue; }
    if (store) {
      if (e->r_epoch == g_epoch && e->r_lane != g_lane && e->r_lane >= 0) report("overwrote", a + i, e->r_lane);
      e->w_epoch = g_epoch; e->w_lane = (int16_t)g_lane;
    } else {
      if (e->w_epoch == g_epoch && e->w_lane != g_lane && e->w_lane >= 0) report("read", a + i, e->w_lane);
      if (e->r_epoch != g_epoch) { e->r_epoch = g_epoch; e->r_lane = (int16_t)g_lane; }      // (the first reader of the epoch is remembered)
    }
  }
}
}  // namespace

extern "C" {
void hipemu_race_init(void) { if (!g_tab) g_tab = (Ent*)calloc(TSIZE, sizeof(Ent)); }
void hipemu_race_exclude(const void* lo, const void* hi) { if (g_nex < 4) { g_ex_lo[g_nex] = (uintptr_t)lo; g_ex_hi[g_nex] = (uintptr_t)hi; g_nex++; } }
void hipemu_race_clear_excludes(void) { g_nex = 0; }
void hipemu_race_lane(int lane) { g_lane = lane; }
void hipemu_race_atomic(int on) { g_atomic = on; }
void hipemu_race_epoch(void) { if (++g_epoch == 0) { g_epoch = 1; if (g_tab) memset(g_tab, 0, TSIZE * sizeof(Ent)); } }
unsigned long hipemu_race_count(void) { return g_races; }
unsigned long hipemu_oob_count(void) { return g_oob; }
void hipemu_race_region(const void* p, size_t n, int add) {
  uintptr_t lo = (uintptr_t)p;
  for (int i = 0; i < g_nreg; i++) if (g_reg[i].lo == lo) { g_reg[i] = g_reg[--g_nreg]; break; }
  if (add && g_nreg < 512) { g_reg[g_nreg].lo = lo; g_reg[g_nreg].hi = lo + n; g_nreg++; }
}
int hipemu_race_sites(void** pcs, unsigned long* counts, int cap) { int k = g_nsites < cap ? g_nsites : cap; for (int i = 0; i < k; i++) { pcs[i] = g_sites[i].pc; counts[i] = g_sites[i].n; } return k; }

#define HOOK(n) \
  void __asan_load##n##_noabort(uintptr_t a) { access(a, n, false); } \
  void __asan_store##n##_noabort(uintptr_t a) { access(a, n, true); }
HOOK(1) HOOK(2) HOOK(4) HOOK(8) HOOK(16)
void __asan_loadN_noabort(uintptr_t a, size_t n) { access(a, n, false); }
void __asan_storeN_noabort(uintptr_t a, size_t n) { access(a, n, true); }
void __asan_handle_no_return(void) {}
void __asan_before_dynamic_init(const char*) {}
void __asan_after_dynamic_init(void) {}
void __asan_init(void) {}
void __asan_version_mismatch_check_v8(void) {}
void __asan_register_globals(void*, uintptr_t) {}
void __asan_unregister_globals(void*, uintptr_t) {}
}
