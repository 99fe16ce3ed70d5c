// Probe of the device call stack on this ROCm: does hipLimitStackSize size the private segment of a kernel
// that recurses (uses_dynamic_stack)?  Every level fills a private frame with a (wave, level) pattern, recurses,
// and verifies the frame afterwards; all waves of the device are resident.  Prints the limits and the number of
// corrupted frames.   hipcc --offload-arch=gfx950 -O2 tools/stack_probe.hip -o /tmp/stack_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__device__ __noinline__ int level(int depth, int maxd, unsigned tag, int* bad) {
  volatile unsigned frame[60];                                  // ~240 B of private stack per level
  for (int i = 0; i < 60; i++) frame[i] = tag * 131u + (unsigned)depth * 7u + (unsigned)i;
  int r = 0;
  if (depth < maxd) r = level(depth + 1, maxd, tag, bad);
  for (int i = 0; i < 60; i++) if (frame[i] != tag * 131u + (unsigned)depth * 7u + (unsigned)i) { atomicAdd(bad, 1); break; }
  return r + (int)frame[depth % 60];
}
__global__ void __launch_bounds__(64, 4) probe(int maxd, int* bad, int* sink) {
  unsigned tag = blockIdx.x * 64u + threadIdx.x;
  int r = level(0, maxd, tag, bad);
  if (r == 0x7fffffff) sink[0] = r;
}
int main(int argc, char** argv) {
  int maxd = argc > 1 ? atoi(argv[1]) : 24;
  size_t want = argc > 2 ? (size_t)atol(argv[2]) : 0;
  size_t lim = 0;
  hipDeviceGetLimit(&lim, hipLimitStackSize); printf("hipLimitStackSize default %zu\n", lim);
  if (want) { hipError_t e = hipDeviceSetLimit(hipLimitStackSize, want); hipDeviceGetLimit(&lim, hipLimitStackSize); printf("set %zu -> %s, now %zu\n", want, hipGetErrorString(e), lim); }
  int *bad, *sink; hipMalloc(&bad, 4); hipMalloc(&sink, 4); hipMemset(bad, 0, 4);
  hipLaunchKernelGGL(probe, dim3(16384), dim3(64), 0, 0, maxd, bad, sink);
  hipError_t e = hipDeviceSynchronize();
  int hb = -1; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
  printf("depth %d (~%d B/lane): sync %s, corrupted frames %d\n", maxd, (maxd + 1) * 260, hipGetErrorString(e), hb);
  return 0;
}
