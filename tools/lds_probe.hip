// Can device code that takes generic (flat) pointers read a block that lives in LDS — unaligned 16-byte loads, byte
// loads — and how much faster is a chain of dependent reads there than in global memory?  (Answer feeds the "small
// blocks are staged in LDS" design: the mutators only ever see `const uint8_t*`.)
//   hipcc --offload-arch=gfx950 -O3 -o build/lds_probe tools/lds_probe.hip && build/lds_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

__device__ __noinline__ uint32_t chase(const uint8_t* p, uint32_t n, uint32_t steps) {   // dependent byte loads
  uint32_t i = threadIdx.x % n, acc = 0;
  for (uint32_t s = 0; s < steps; s++) { uint32_t b = p[i]; acc += b; i = (i * 7u + b + 1u) % n; }
  return acc;
}
__device__ __noinline__ uint32_t sum16(const uint8_t* p, uint32_t n, uint32_t shift) {   // unaligned 16-byte vector loads
  uint32_t acc = 0;
  for (uint32_t i = shift + 16u * threadIdx.x; i + 16 <= n; i += 1024) { uint4 v; __builtin_memcpy(&v, p + i, 16); acc += v.x ^ v.y ^ v.z ^ v.w; }
  return acc;
}
__global__ void __launch_bounds__(64) probe(const uint8_t* g, uint32_t n, uint32_t steps, uint32_t* out, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[8192];
  for (uint32_t i = threadIdx.x; i < n; i += 64) lds[i] = g[i];
  __syncthreads();
  const uint8_t* lp = lds;                      // generic pointer into the LDS aperture
  unsigned long long t0 = __builtin_readcyclecounter();
  uint32_t a = chase(g, n, steps);
  unsigned long long t1 = __builtin_readcyclecounter();
  uint32_t b = chase(lp, n, steps);
  unsigned long long t2 = __builtin_readcyclecounter();
  uint32_t c = 0, d = 0;
  for (uint32_t sh = 0; sh < 16; sh++) { c += sum16(g, n, sh); }
  unsigned long long t3 = __builtin_readcyclecounter();
  for (uint32_t sh = 0; sh < 16; sh++) { d += sum16(lp, n, sh); }
  unsigned long long t4 = __builtin_readcyclecounter();
  out[4 * threadIdx.x] = a; out[4 * threadIdx.x + 1] = b; out[4 * threadIdx.x + 2] = c; out[4 * threadIdx.x + 3] = d;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; }
}
int main() {
  const uint32_t n = 4096, steps = 2000;
  std::vector<uint8_t> h(n); for (uint32_t i = 0; i < n; i++) h[i] = (uint8_t)(i * 131u + (i >> 3));
  uint8_t* d; uint32_t* o; unsigned long long* c;
  hipMalloc(&d, n); hipMalloc(&o, 64 * 16); hipMalloc(&c, 32);
  hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, n, steps, o, c);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  uint32_t ho[256]; unsigned long long hc[4];
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost); hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
  int bad = 0; for (int l = 0; l < 64; l++) { if (ho[4 * l] != ho[4 * l + 1]) bad++; if (ho[4 * l + 2] != ho[4 * l + 3]) bad++; }
  printf("mismatches %d; dependent byte loads: global %.0f cycles/step, LDS via flat %.0f cycles/step; unaligned 16-byte sweeps of 4 KiB: global %llu, LDS %llu cycles\n",
         bad, (double)hc[0] / steps, (double)hc[1] / steps, hc[2] / 16, hc[3] / 16);
  return bad != 0;
}
