#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(long* o, float a, float b, int n) { float x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3; long t0 = clock64(); long w0 = wall_clock64();
  for (int i = 0; i < n; ++i) { x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b); asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); }
  long t1 = clock64(); long w1 = wall_clock64(); if (threadIdx.x % 64 == 0) { o[0 + 4 * (threadIdx.x / 64)] = t1 - t0; o[1 + 4 * (threadIdx.x / 64)] = w1 - w0; o[2] = (long)(x0 + x1 + x2 + x3); } }
int main() { long* o; hipMalloc(&o, 4096); long h[64]; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nt : {64, 256, 512, 768}) for (int rep = 0; rep < 2; ++rep) { int n = 4000000; hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(1), dim3(nt), 0, 0, o, 1.0000001f, 1e-9f, n); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
    printf("threads %d: %.2f ms  clock64 ticks %ld (%.3f GHz)  wall_clock64 %ld (%.1f MHz)  ticks per fma-instr per wave %.2f  => SIMD issues one wave-fma per %.2f ticks\n", nt, ms, h[0], h[0] / (ms * 1e6), h[1], h[1] / (ms * 1e3), (double)h[0] / (4.0 * n), (double)h[0] / (4.0 * n) / ((nt + 255) / 256)); }
  return 0; }
