// micro-benchmarks of the gfx950 properties the decode kernel depends on (diagnostic tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
__global__ void k_dep(float* o, float a, float b) { float x = threadIdx.x; long t0 = clock64();
  for (int i = 0; i < N; ++i) x = fmaf(x, a, b);
  long t1 = clock64(); if (threadIdx.x == 0) { o[blockIdx.x * 4] = x; ((long*)o)[1 + blockIdx.x] = t1 - t0; } }
__global__ void k_ind(float* o, float a, float b) { float x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7; long t0 = clock64();
  for (int i = 0; i < N / 8; ++i) { x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b); x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b); }
  long t1 = clock64(); if (threadIdx.x % 64 == 0) { o[0] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7; ((long*)o)[1 + threadIdx.x / 64] = t1 - t0; } }
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k_pk(float* o, float a, float b) { f2 x0 = {(float)threadIdx.x, 1}, x1 = {2, 3}, x2 = {4, 5}, x3 = {6, 7}; f2 A = {a, a}, B = {b, b}; long t0 = clock64();
  for (int i = 0; i < N / 4; ++i) { x0 = __builtin_elementwise_fma(x0, A, B); x1 = __builtin_elementwise_fma(x1, A, B); x2 = __builtin_elementwise_fma(x2, A, B); x3 = __builtin_elementwise_fma(x3, A, B); }
  long t1 = clock64(); if (threadIdx.x % 64 == 0) { o[0] = x0.x + x1.y + x2.x + x3.y; ((long*)o)[1 + threadIdx.x / 64] = t1 - t0; } }
__global__ void k_bar(float* o) { __shared__ float s[1024]; long t0 = clock64();
  for (int i = 0; i < 1024; ++i) { s[threadIdx.x] = i; __syncthreads(); }
  long t1 = clock64(); if (threadIdx.x == 0) { o[0] = s[5]; ((long*)o)[1] = t1 - t0; } }
__global__ void k_handoff(float* o) { __shared__ float s[1024]; float v = threadIdx.x; long t0 = clock64();
  for (int i = 0; i < 1024; ++i) { s[threadIdx.x] = v; __syncthreads(); v = s[(threadIdx.x + 64) % blockDim.x] + 1.0f; __syncthreads(); }
  long t1 = clock64(); if (threadIdx.x == 0) { o[0] = v; ((long*)o)[1] = t1 - t0; } }
__global__ void k_lds(float* o) { __shared__ int s[1024]; for (int i = threadIdx.x; i < 1024; i += blockDim.x) s[i] = (i * 7 + 1) % 1024; __syncthreads(); int p = threadIdx.x; long t0 = clock64();
  for (int i = 0; i < 1024; ++i) p = s[p];
  long t1 = clock64(); if (threadIdx.x == 0) { o[0] = p; ((long*)o)[1] = t1 - t0; } }
__global__ void k_gld(const int* g, float* o) { int p = threadIdx.x; long t0 = clock64();
  for (int i = 0; i < 256; ++i) p = g[p];
  long t1 = clock64(); if (threadIdx.x == 0) { o[0] = p; ((long*)o)[1] = t1 - t0; } }
int main() { float* o; hipMalloc(&o, 1 << 16); long h[64]; int* g; hipMalloc(&g, 64 << 20);
  { int n = 16 << 20; int* hg = (int*)malloc(n * 4); for (int i = 0; i < n; ++i) hg[i] = (int)(((long)i * 1315423911u + 12345) % n); hipMemcpy(g, hg, n * 4, hipMemcpyHostToDevice); }
  auto rd = [&]() { hipDeviceSynchronize(); hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost); };
  for (int rep = 0; rep < 2; ++rep) {
  hipLaunchKernelGGL(k_dep, dim3(1), dim3(64), 0, 0, o, 1.0001f, 0.5f); rd(); printf("dependent fma, 1 wave:        %.2f cyc/instr\n", (double)h[1] / N);
  hipLaunchKernelGGL(k_ind, dim3(1), dim3(64), 0, 0, o, 1.0001f, 0.5f); rd(); printf("independent fma x8, 1 wave:   %.2f cyc/instr\n", (double)h[1] / N);
  hipLaunchKernelGGL(k_ind, dim3(1), dim3(256), 0, 0, o, 1.0001f, 0.5f); rd(); printf("independent fma, 1 wave/SIMD:  %.2f cyc/instr/wave\n", (double)h[1] / N);
  hipLaunchKernelGGL(k_ind, dim3(1), dim3(512), 0, 0, o, 1.0001f, 0.5f); rd(); printf("independent fma, 2 waves/SIMD: %.2f cyc/instr/wave\n", (double)h[1] / N);
  hipLaunchKernelGGL(k_ind, dim3(1), dim3(768), 0, 0, o, 1.0001f, 0.5f); rd(); printf("independent fma, 3 waves/SIMD: %.2f cyc/instr/wave\n", (double)h[1] / N);
  hipLaunchKernelGGL(k_pk, dim3(1), dim3(64), 0, 0, o, 1.0001f, 0.5f); rd(); printf("independent pk_fma x4, 1 wave: %.2f cyc/instr\n", (double)h[1] / N);
  hipLaunchKernelGGL(k_pk, dim3(1), dim3(768), 0, 0, o, 1.0001f, 0.5f); rd(); printf("independent pk_fma, 3 w/SIMD:  %.2f cyc/instr/wave\n", (double)h[1] / N);
  hipLaunchKernelGGL(k_bar, dim3(1), dim3(768), 0, 0, o); rd(); printf("ds_write + __syncthreads, 12 waves: %.1f cyc/iter\n", (double)h[1] / 1024);
  hipLaunchKernelGGL(k_handoff, dim3(1), dim3(768), 0, 0, o); rd(); printf("write->barrier->read->barrier, 12 waves: %.1f cyc/iter\n", (double)h[1] / 1024);
  hipLaunchKernelGGL(k_handoff, dim3(1), dim3(256), 0, 0, o); rd(); printf("write->barrier->read->barrier, 4 waves: %.1f cyc/iter\n", (double)h[1] / 1024);
  hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 0, 0, o); rd(); printf("dependent ds_read_b32: %.1f cyc\n", (double)h[1] / 1024);
  hipLaunchKernelGGL(k_gld, dim3(1), dim3(64), 0, 0, g, o); rd(); printf("dependent global_load (64MB table, per-lane random): %.1f cyc\n", (double)h[1] / 256);
  }
  return 0; }
